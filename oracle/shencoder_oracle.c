/* oracle/shencoder_oracle.c -- TEST INFRASTRUCTURE (CPU oracle), NOT product code.
 *
 * Real spherical-harmonics direction encoding of degree C <= 8 (C*C outputs) and its analytic
 * Jacobian, restating what the reference's kernel_sh / kernel_sh_backward compute
 * (reference: lib/ops/shencoder/src/shencoder.cu:27-356 and :359-383).
 *
 * The reference spells every basis function out as a polynomial in (x, y, z) with the unit-sphere
 * identity folded in.  This oracle states the same functions in closed form instead:
 *
 *     Y[l*l + l + m] = (-1)^m * s_m * K(l,|m|) * Q(l,|m|; z) * (m >= 0 ? A_|m|(x,y) : B_|m|(x,y))
 *
 *   K(l,m)   = sqrt((2l+1)/(4 pi) * (l-m)!/(l+m)!),      s_0 = 1, s_m = sqrt(2) for m != 0
 *   Q(l,m;z) = d^m/dz^m P_l(z)   (m-th derivative of the Legendre polynomial; polynomial in z)
 *   A_m + i B_m = (x + i y)^m
 *
 * and dY/dx, dY/dy, dY/dz follow from dA_m/dx = m A_{m-1}, dA_m/dy = -m B_{m-1},
 * dB_m/dx = m B_{m-1}, dB_m/dy = m A_{m-1}, dQ(l,m)/dz = Q(l,m+1) - which is exactly the partial
 * derivative of the reference's polynomials (shencoder.cu:131-351).  Evaluated in double, rounded
 * to fp32 once; the reference evaluates in fp32, so comparisons use a tolerance (tests state it).
 * Pinned against oracle/_ref (the reference's kernel compiled for the CPU) for all 64 outputs and
 * all 192 Jacobian entries in tests/test_oracle_vs_reference.py.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define ORC_SH_MAXDEG 8

static double orc_fact(int n) { double f = 1.0; for (int i = 2; i <= n; ++i) f *= i; return f; }

/* Q[l][m] for 0<=m<=l<C plus Q[l][l+1]=0, via the Legendre derivative recurrence
 * Q(m,m) = (2m-1)!!,  Q(m+1,m) = (2m+1) z Q(m,m),  (l-m) Q(l,m) = (2l-1) z Q(l-1,m) - (l+m-1) Q(l-2,m) */
static void orc_legendre_derivs(int C, double z, double Q[ORC_SH_MAXDEG][ORC_SH_MAXDEG + 2]) {
    for (int l = 0; l < C; ++l) for (int m = 0; m < ORC_SH_MAXDEG + 2; ++m) Q[l][m] = 0.0;
    for (int m = 0; m < C; ++m) {
        double dfact = 1.0;
        for (int k = 2 * m - 1; k > 1; k -= 2) dfact *= k;
        Q[m][m] = dfact;
        if (m + 1 < C) Q[m + 1][m] = (2 * m + 1) * z * Q[m][m];
        for (int l = m + 2; l < C; ++l)
            Q[l][m] = ((2 * l - 1) * z * Q[l - 1][m] - (l + m - 1) * Q[l - 2][m]) / (double)(l - m);
    }
}

void orc_sh_encode_forward(const float* inputs, float* outputs, uint32_t B, uint32_t D, uint32_t C, int calc_grad_inputs,
                           float* dy_dx) {
    const double PI = 3.14159265358979323846;
    const uint32_t C2 = C * C;
    for (uint32_t b = 0; b < B; ++b) {
        const double x = inputs[(size_t)b * D + 0], y = inputs[(size_t)b * D + 1], z = inputs[(size_t)b * D + 2];
        double Q[ORC_SH_MAXDEG][ORC_SH_MAXDEG + 2];
        orc_legendre_derivs((int)C, z, Q);
        double A[ORC_SH_MAXDEG + 1], Bm[ORC_SH_MAXDEG + 1];
        A[0] = 1.0; Bm[0] = 0.0;
        for (uint32_t m = 1; m <= C; ++m) { A[m] = x * A[m - 1] - y * Bm[m - 1]; Bm[m] = x * Bm[m - 1] + y * A[m - 1]; }
        float* out = outputs + (size_t)b * C2;
        float* gx = calc_grad_inputs ? dy_dx + (size_t)b * D * C2 : 0;
        float* gy = gx ? gx + C2 : 0;
        float* gz = gy ? gy + C2 : 0;
        for (int l = 0; l < (int)C; ++l) {
            for (int m = -l; m <= l; ++m) {
                const int am = m < 0 ? -m : m;
                const double K = sqrt((2.0 * l + 1.0) / (4.0 * PI) * orc_fact(l - am) / orc_fact(l + am));
                const double s = (am == 0 ? 1.0 : sqrt(2.0)) * ((am & 1) ? -1.0 : 1.0);
                const double xy = m >= 0 ? A[am] : Bm[am];
                const int idx = l * l + l + m;
                out[idx] = (float)(s * K * Q[l][am] * xy);
                if (gx) {
                    double dxy_dx = 0.0, dxy_dy = 0.0;
                    if (am > 0) {
                        dxy_dx = m >= 0 ? am * A[am - 1] : am * Bm[am - 1];
                        dxy_dy = m >= 0 ? -am * Bm[am - 1] : am * A[am - 1];
                    }
                    gx[idx] = (float)(s * K * Q[l][am] * dxy_dx);
                    gy[idx] = (float)(s * K * Q[l][am] * dxy_dy);
                    gz[idx] = (float)(s * K * Q[l][am + 1 <= l ? am + 1 : ORC_SH_MAXDEG + 1] * xy);
                }
            }
        }
    }
}

/* grad_inputs[b,d] += sum_ch grad[b,ch] * dy_dx[b,d,ch]   (shencoder.cu:359-383; accumulates in place) */
void orc_sh_encode_backward(const float* grad, const float* inputs, uint32_t B, uint32_t D, uint32_t C, const float* dy_dx,
                            float* grad_inputs) {
    (void)inputs;
    const uint32_t C2 = C * C;
    for (uint32_t b = 0; b < B; ++b)
        for (uint32_t d = 0; d < D; ++d) {
            float acc = grad_inputs[(size_t)b * D + d];
            for (uint32_t ch = 0; ch < C2; ++ch) acc = fmaf(grad[(size_t)b * C2 + ch], dy_dx[((size_t)b * D + d) * C2 + ch], acc);
            grad_inputs[(size_t)b * D + d] = acc;
        }
}
