#!/usr/bin/env python3
"""Builds dark_amd/libdark_hip_prof.so with attn_kernel writing wall_clock64 timestamps at its phase boundaries
(thread 0 of each workgroup) plus the reader `dark_dbg_attn_ts`; profiling aid for tools/probe/attn_prof.py:
    python tools/probe/make_attn_prof.py
    gpurun -- 'cp dark_amd/libdark_hip_prof.so dark_amd/libdark_hip.so && python3 tools/probe/attn_prof.py'"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dark_amd", "csrc")
src = open(os.path.join(CSRC, "attn.hip")).read()


def rep(old, new):
    global src
    assert src.count(old) == 1, (src.count(old), old)
    src = src.replace(old, new)


rep('template <typename T, int HD, int NT, int NW>\n__global__ __launch_bounds__(NW * 64) void attn_kernel(',
    '__device__ unsigned long long g_ats[8192 * 8];\n'
    '#define PROF(k) do { if (tid == 0 && blockIdx.x < 8192) g_ats[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)\n'
    'template <typename T, int HD, int NT, int NW>\n__global__ __launch_bounds__(NW * 64) void attn_kernel(')
rep('  // ---- stage Q, K with RoPE\n', '  PROF(0);\n  // ---- stage Q, K with RoPE\n')
rep('  for (int j = tid; j < NP; j += NTH) {\n    const bool on = j < N && (!kmask',
    '  PROF(1);\n  for (int j = tid; j < NP; j += NTH) {\n    const bool on = j < N && (!kmask')
rep('  __syncthreads();\n\n  const int li = lane & 15, g = lane >> 4;', '  __syncthreads();\n  PROF(2);\n\n  const int li = lane & 15, g = lane >> 4;')
rep('    // ---- softmax over keys for query qi', '    PROF(3);\n    // ---- softmax over keys for query qi')
rep('    // ---- O = P V\n', '    PROF(4);\n    // ---- O = P V\n')
rep('          if (i < N) orow[(int64_t)i * dm + dt * 16 + li] = (bf16_t)acc[r];\n        }\n      }\n    }\n  }\n}',
    '          if (i < N) orow[(int64_t)i * dm + dt * 16 + li] = (bf16_t)acc[r];\n        }\n      }\n    }\n    PROF(5);\n  }\n}')
rep('}  // namespace dark', '}  // namespace dark\nextern "C" int dark_dbg_attn_ts(unsigned long long* out, int n) {\n'
    '  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dark::g_ats), sizeof(unsigned long long) * n) != hipSuccess;\n}\n')
os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
open(os.path.join(CSRC, "build", "attn_prof.hip"), "w").write(src)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I.", "-I../../include", "-c",
                       "build/attn_prof.hip", "-o", "build/attn_prof.o"], cwd=CSRC)
objs = [f"build/{n}.o" for n in ("api", "gemm", "elementwise", "sample", "expand", "grpo", "layout")] + ["build/attn_prof.o"]
subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", "../libdark_hip_prof.so"], cwd=CSRC)
print("built dark_amd/libdark_hip_prof.so")
