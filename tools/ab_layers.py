#!/usr/bin/env python
"""tools/ab_layers.py A.jsonl B.jsonl -- per-layer comparison of two tools/bench_conv.py logs (same box): flags layers that differ by more than 4 %"""
import json, sys
P = [json.loads(l) for l in open(sys.argv[1]) if l.startswith('{"H"')]
N = [json.loads(l) for l in open(sys.argv[2]) if l.startswith('{"H"')]
tp = tn = 0.0
for n in N:
    p = [q for q in P if all(q[k] == n[k] for k in ("H", "Cin", "Cout", "k", "stride", "up"))]
    pu, nu = (p[0]["own_us"]["0"] if p else float("nan")), n["own_us"]["0"]
    if p: tp += pu * n["n"]; tn += nu * n["n"]
    flag = " <<<" if p and nu > pu * 1.04 else " >>>" if p and nu < pu * 0.96 else ""
    print(f"  H{n['H']:4d} {n['Cin']:5d}->{n['Cout']:4d} k{n['k']} s{n['stride']} up{n['up']} x{n['n']:2d}  A {pu:8.1f}  B {nu:8.1f}{flag}")
print(f"  common layers, count-weighted: A {tp / 1e3:.3f} ms  B {tn / 1e3:.3f} ms")
