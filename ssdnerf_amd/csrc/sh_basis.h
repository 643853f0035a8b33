// ssdnerf_amd/csrc/sh_basis.h -- real spherical-harmonics basis (degree <= 8) for gfx950, in registers.
//
// Same functions as the reference's kernel_sh (lib/ops/shencoder/src/shencoder.cu:44-121), stated as
//   Y[l*l+l+m] = c(l,m) * Q(l,|m|; z) * (m >= 0 ? A_|m|(x,y) : B_|m|(x,y)),
//   c(l,m) = (-1)^m * (m ? sqrt2 : 1) * sqrt((2l+1)/(4pi) * (l-|m|)!/(l+|m|)!)
//   Q(l,m;z) = d^m/dz^m P_l(z)  (recurrence below),  A_m + iB_m = (x+iy)^m,
// with every loop bound a template constant so the compiler emits straight-line fp32 code with the
// normalisation constants folded.  The Jacobian uses dA_m/dx = mA_{m-1}, dA_m/dy = -mB_{m-1},
// dB_m/dx = mB_{m-1}, dB_m/dy = mA_{m-1}, dQ(l,m)/dz = Q(l,m+1)  (== shencoder.cu:131-351).
#pragma once
#include "common.h"

namespace shb {

constexpr double kPi = 3.14159265358979323846;
constexpr double cfact(int n) { double f = 1.0; for (int i = 2; i <= n; ++i) f *= i; return f; }
constexpr double csqrt(double x) {  // Newton iterations; constexpr-friendly
    double r = x > 1.0 ? x : 1.0;
    for (int i = 0; i < 64; ++i) r = 0.5 * (r + x / r);
    return r;
}
constexpr float norm_const(int l, int m) {
    const int am = m < 0 ? -m : m;
    const double k = csqrt((2.0 * l + 1.0) / (4.0 * kPi) * cfact(l - am) / cfact(l + am));
    const double s = (am == 0 ? 1.0 : csqrt(2.0)) * ((am & 1) ? -1.0 : 1.0);
    return (float)(s * k);
}
constexpr float dfact_odd(int m) {  // (2m-1)!!
    double f = 1.0;
    for (int k = 2 * m - 1; k > 1; k -= 2) f *= k;
    return (float)f;
}

// All constants are materialised at COMPILE time in one table (a plain call to the constexpr functions above from
// device code is evaluated at run time - in fp64, with a 64-step Newton sqrt - unless forced into a constant expression).
struct Tables {
    float norm[64];   // norm[l*l + l + m] = c(l, m)
    float dfact[9];   // (2m-1)!!
};
constexpr Tables make_tables() {
    Tables t{};
    for (int l = 0; l < 8; ++l)
        for (int m = -l; m <= l; ++m) t.norm[l * l + l + m] = norm_const(l, m);
    for (int m = 0; m < 9; ++m) t.dfact[m] = dfact_odd(m);
    return t;
}

constexpr float C0 = make_tables().norm[0];   // SH_0 of every direction (exactly what eval() returns in out[0])

// Evaluates the C*C basis values (and optionally the three Jacobian rows) of one direction.
template <int C, bool GRAD>
SSD_DEV void eval(float x, float y, float z, float* __restrict__ out, float* __restrict__ gx, float* __restrict__ gy, float* __restrict__ gz) {
    constexpr Tables T = make_tables();
    float A[C + 1], B[C + 1];
    A[0] = 1.0f; B[0] = 0.0f;
#pragma unroll
    for (int m = 1; m <= C; ++m) {
        A[m] = ssd_fma(x, A[m - 1], -(y * B[m - 1]));
        B[m] = ssd_fma(x, B[m - 1], y * A[m - 1]);
    }
    float Q[C][C + 1];
#pragma unroll
    for (int l = 0; l < C; ++l)
#pragma unroll
        for (int m = 0; m <= C; ++m) Q[l][m] = 0.0f;
#pragma unroll
    for (int m = 0; m < C; ++m) {
        Q[m][m] = T.dfact[m];
        if (m + 1 < C) Q[m + 1][m] = (float)(2 * m + 1) * z * Q[m][m];
#pragma unroll
        for (int l = m + 2; l < C; ++l)
            Q[l][m] = ssd_fma((float)(2 * l - 1) * z, Q[l - 1][m], -(float)(l + m - 1) * Q[l - 2][m]) * (1.0f / (float)(l - m));
    }
#pragma unroll
    for (int l = 0; l < C; ++l) {
#pragma unroll
        for (int m = -l; m <= l; ++m) {
            const int am = m < 0 ? -m : m;
            const float c = T.norm[l * l + l + m];
            const float xy = m >= 0 ? A[am] : B[am];
            const int idx = l * l + l + m;
            out[idx] = c * Q[l][am] * xy;
            if (GRAD) {
                float dx = 0.0f, dy = 0.0f;
                if (am > 0) {
                    dx = (float)am * (m >= 0 ? A[am - 1] : B[am - 1]);
                    dy = (float)am * (m >= 0 ? -B[am - 1] : A[am - 1]);
                }
                gx[idx] = c * Q[l][am] * dx;
                gy[idx] = c * Q[l][am] * dy;
                gz[idx] = c * Q[l][am + 1] * xy;  // Q[l][l+1] == 0
            }
        }
    }
}

}  // namespace shb
