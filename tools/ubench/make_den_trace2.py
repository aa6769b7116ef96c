"""Per-block phase timeline of density_lr_kernel WITHOUT disturbing its register allocation (tools/ubench/den_trace.hip's -DDEN_TRACE
build spills 273 VGPRs): a patched copy of csrc/grid_density.hip keeps four s_memrealtime stamps in SGPRs (start, end of phase 1, end
of the last panel's phase 2, end) and thread 0 writes them with the CU id when the block is done.
usage: python tools/ubench/make_den_trace2.py  -> tools/ubench/_den_trace2_kernel.hip (included by den_trace2.hip)
(DEN_TRACE_STEPS=1 adds per-step stamps to the experimental direct-operand phase 1 of docs/LOG_r06.md section 4; that code path is not in
the product kernel, so the option only applies to a tree that carries it.  den_trace2 <us> delays the first blocks in odd wave slots.)"""
import os
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "../../dqc_amd/csrc/grid_density.hip")).read()
i0 = src.index("template <int NRT, int NCT, bool GGA, int NS = 1>\n__global__ __launch_bounds__(256, 2) void density_lr_kernel(")
head, body = src[:i0], src[i0:]
end = body.index("\ntemplate <int NRT, int NCT>\nstatic constexpr size_t density_lr_lds_bytes()")
kern, rest = body[:end], body[end:]


def rep(s, a, b):
    assert a in s, a[:50]
    return s.replace(a, b, 1)


kern = rep(kern, "    const int g0 = blockIdx.x * DEN_BM;\n", """    if (blockIdx.x < 512 && g_den_stagger > 0) {  // optional start offset of the first blocks in odd wave slots (argv[1], us)
        unsigned hw0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw0));
        if (hw0 & 1u) {
            const long long t0_ = wall_clock64();
            while (wall_clock64() - t0_ < g_den_stagger) __builtin_amdgcn_s_sleep(32);
        }
    }
    const long long tr0_ = wall_clock64();
    long long tr2_ = 0;
    const int g0 = blockIdx.x * DEN_BM;
""")
kern = rep(kern, "    // rho_g = sum_r A'[g][r]^2 straight from the phase-1 accumulators", "    const long long tr1_ = wall_clock64();\n    // rho_g = sum_r A'[g][r]^2 straight from the phase-1 accumulators")
kern = rep(kern, "        DEN_TRACE_POINT(1);\n", "        tr2_ = wall_clock64();\n")
# optional per-step stamps of the direct phase 1 (env DEN_TRACE_STEPS=1): after the first staging barrier and after every step
if os.environ.get("DEN_TRACE_STEPS"):
    kern = rep(kern, "            DQC_P1_LPUT(0, 0)\n            __syncthreads();\n", "            DQC_P1_LPUT(0, 0)\n            __syncthreads();\n            long long ts_[9]; ts_[0] = wall_clock64();\n")
    kern = rep(kern, "                if (s_ + 1 < NSTEP) { DQC_P1_LPUT((s_ + 1) & 1, s_ + 1) }\n                __syncthreads();\n", "                if (s_ + 1 < NSTEP) { DQC_P1_LPUT((s_ + 1) & 1, s_ + 1) }\n                __syncthreads();\n                if (s_ + 1 < 9) ts_[s_ + 1] = wall_clock64();\n")
    kern = rep(kern, "#undef DQC_P1_PHI\n", "            if (threadIdx.x == 0 && blockIdx.x < 16384) { for (int q = 0; q < 8; q++) g_den_steps[8 * (size_t)blockIdx.x + q] = (q <= NSTEP ? ts_[q] : ts_[NSTEP]) - tr0_; }\n#undef DQC_P1_PHI\n")
    head = head + "__device__ long long g_den_steps[8 * 16384];\n"
# the kernel's closing brace: append the write-out before it
k = kern.rstrip()
assert k.endswith("}")
k = k[:-1] + """    {
        const long long tr3_ = wall_clock64();
        if (threadIdx.x == 0 && blockIdx.x < 16384) {
            unsigned hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            long long *o = g_den_trace2 + 5 * (size_t)blockIdx.x;
            o[0] = tr0_; o[1] = tr1_; o[2] = tr2_; o[3] = tr3_;
            o[4] = (long long)(((xcc & 7u) << 12) | (hw & 0xfffu));
        }
    }
}
"""
head = head + "__device__ long long g_den_trace2[5 * 16384];\n__device__ int g_den_stagger = 0;\n"
open(os.path.join(here, "_den_trace2_kernel.hip"), "w").write(head + k + rest)
print("written")
