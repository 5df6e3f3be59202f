#!/usr/bin/env python3
"""On the GPU box: the measured vector-instruction issue rates (cgpt_measure_issue_rate) for every kind at 1..8 waves per SIMD."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cpugpupathtracing_amd as P
r = P.Renderer(0)
names = ["v_mul_f32", "v_pk_mul_f32", "v_pk_add_f32", "v_rcp_f32", "3 v_mul : 1 v_pk_mul interleaved", "48 v_mul + 16 v_pk_mul grouped",
         "1 v_mul : 1 v_pk_mul alternating", "v_cndmask_b32 (vcc)", "v_mul_lo_u32", "v_cndmask_b32_e64 (sgpr pair)",
         "v_cmp_lt_f32 + v_cndmask_b32 pairs", "v_add_u32", "v_min3_f32"]
print("Gwave-inst/s over the chip (256 CUs x 4 SIMDs), by resident waves per SIMD 1..8; fastest of three launches of 60000 x 64 instructions per wave")
for kind, nm in enumerate(names):
    row = [r.measure_issue_rate(kind=kind, waves_per_simd=w, iters=60000)[0] / 1e9 for w in range(1, 9)]
    print(f"{kind:2d} {nm:36s} " + " ".join(f"{v:7.1f}" for v in row), flush=True)
