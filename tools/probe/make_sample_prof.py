#!/usr/bin/env python3
"""Builds dark_amd/libdark_hip_prof.so: the library with sample_kernel writing wall_clock64 timestamps at its phase
boundaries (thread 0 of each workgroup, plain stores into a static device array) plus a reader `dark_dbg_sample_ts`.
Profiling aid only -- the product library never contains this.  Use:
    python tools/probe/make_sample_prof.py
    gpurun -- 'cp dark_amd/libdark_hip_prof.so dark_amd/libdark_hip.so && python3 tools/probe/sample_prof.py'
(the copy happens on the GPU box's scratch tree only)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "dark_amd", "csrc")
src = open(os.path.join(CSRC, "sample.hip")).read()


def rep(old, new):
    global src
    assert src.count(old) == 1, (src.count(old), old)
    src = src.replace(old, new)


rep('\ntemplate <bool GROW>\n__device__ __forceinline__ void sample_row(',
    '\n__device__ unsigned long long g_ts[32768 * 10];\n'
    '#define PROF(k) do { if (tid == 0 && m < 32768) g_ts[m * 10 + (k)] = wall_clock64(); } while (0)\n'
    'template <bool GROW>\n__device__ __forceinline__ void sample_row(')
rep('  const float* lr = logits + (int64_t)blockIdx.x * ldl;\n\n  // 1. temperature', '  const float* lr = logits + (int64_t)blockIdx.x * ldl;\n  PROF(0);\n\n  // 1. temperature')
rep('  if (flat_hint) block_max2(mx, nmn, s, tid);  // (block-uniform)\n  else mx = block_max(mx, s, tid);\n\n  int list_c',
    '  PROF(1);\n  if (flat_hint) block_max2(mx, nmn, s, tid);\n  else mx = block_max(mx, s, tid);\n  PROF(2);\n\n  int list_c')
rep('    if (rc == 2) {', '    PROF(3);\n    if (rc == 2) {')
rep('                       (unsigned long long)m * V, tid, conf + m, x0 + m);\n      return;',
    '                       (unsigned long long)m * V, tid, conf + m, x0 + m);\n'
    '      for (int k_ = 4; k_ <= 8; ++k_) PROF(k_);\n      return;')
rep('  // 3. top-k filter', '  PROF(4);\n  // 3. top-k filter')
rep('  const float Zf = (float)block_sum(zs, s, tid);\n', '  const float Zf = (float)block_sum(zs, s, tid);\n  PROF(5);\n')
rep('  const float psum = (float)ps;\n  const float entropy = (float)ent;\n\n  // 5. draw',
    '  const float psum = (float)ps;\n  const float entropy = (float)ent;\n  PROF(6);\n\n  // 5. draw')
rep('  // block arg-max, ties -> lowest index; merge', '  PROF(7);\n  // block arg-max, ties -> lowest index; merge')
rep('    s->bin = bi;\n  }\n  if (list && conf_mode', '    s->bin = bi;\n  }\n  PROF(8);\n  if (list && conf_mode')
# stamps inside nucleus_crossing (g_tx, read by dark_dbg_sample_tx): entry, [flat: zeroed, fused mass+histogram pass,
# block sum | else: mass pass, block sum, histogram pass], scan, crossing-bin gather, resolve
rep('\n__device__ int nucleus_crossing(',
    '\n__device__ unsigned long long g_tx[32768 * 8];\n'
    '#define PROFX(k) do { if (tid == 0 && blockIdx.x < 32768) g_tx[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)\n'
    '__device__ int nucleus_crossing(')
rep('  *head_c = -1;\n  unsigned long long T;\n', '  *head_c = -1;\n  unsigned long long T;\n  PROFX(0);\n')
rep('    unsigned long long zl = 0;\n    const int v4 = V >> 2;\n', '    PROFX(1);\n    unsigned long long zl = 0;\n    const int v4 = V >> 2;\n')
rep('    zl = block_sum_u64(zl, s, tid);  // (its barriers also publish the histogram)\n',
    '    PROFX(2);\n    zl = block_sum_u64(zl, s, tid);\n    PROFX(3);\n')
rep('  block_sum2_u64(zl, hl, s, tid);  // (its barriers also publish the candidates)\n',
    '  PROFX(1);\n  block_sum2_u64(zl, hl, s, tid);\n  PROFX(2);\n')
rep('  __syncthreads();\n  }  // !flat_hint\n', '  __syncthreads();\n  PROFX(3);\n  }  // !flat_hint\n')
rep('  if (!s->found) return 0;\n  const int b = s->bin;', '  PROFX(4);\n  if (!s->found) return 0;\n  const int b = s->bin;')
rep('  const int c = s->ncand;\n  if (c > CAND_CAP) return -1;\n  return resolve_candidates(cand, cfx, c, A, T, s, tid, tau, above, n_tau);',
    '  const int c = s->ncand;\n  PROFX(5);\n  if (c > CAND_CAP) return -1;\n'
    '  const int rr_ = resolve_candidates(cand, cfx, c, A, T, s, tid, tau, above, n_tau);\n  PROFX(6);\n  return rr_;')
rep('}  // namespace dark\n', '}  // namespace dark\nextern "C" int dark_dbg_sample_tx(unsigned long long* out, int n) {\n'
    '  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dark::g_tx), sizeof(unsigned long long) * n) != hipSuccess;\n}\n'
    'extern "C" int dark_dbg_sample_ts(unsigned long long* out, int n) {\n'
    '  return hipMemcpyFromSymbol(out, HIP_SYMBOL(dark::g_ts), sizeof(unsigned long long) * n) != hipSuccess;\n}\n')
os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
open(os.path.join(CSRC, "build", "sample_prof.hip"), "w").write(src)
hipcc = "/opt/rocm/bin/hipcc"
subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I.", "-I../../include", "-c",
                       "build/sample_prof.hip", "-o", "build/sample_prof.o"], cwd=CSRC)
objs = [f"build/{n}.o" for n in ("api", "gemm", "gemm256", "gemm_rowpanel", "attn", "elementwise", "expand", "grpo", "layout", "gpt2")] + ["build/sample_prof.o"]
subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", "../libdark_hip_prof.so"], cwd=CSRC)
print("built dark_amd/libdark_hip_prof.so")
