// Issue-rate microbenchmark for gfx950: cycles per instruction of one wave's stream at 1..8 waves per SIMD, for a few
// instruction mixes.  hipcc --offload-arch=gfx950 -O3 -o issue issue.hip && ./issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MIX>
__global__ void k(float* out, unsigned long long* cyc, int iters, unsigned long long seed) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, b = 1.0001f;
  unsigned long long m = seed | 1ull;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (MIX == 0) {   // 16 independent-ish VALU adds (4 chains)
#pragma unroll
      for (int j = 0; j < 4; ++j) { a0 += b; a1 += b; a2 += b; a3 += b; }
    } else if (MIX == 1) {   // 16 dependent VALU adds (1 chain)
#pragma unroll
      for (int j = 0; j < 16; ++j) a0 += b;
    } else if (MIX == 2) {   // 8 x (scalar shift + masked select + add), one chain
#pragma unroll
      for (int j = 0; j < 8; ++j) { m = (m << 1) | (m >> 63); a0 += __builtin_amdgcn_inverse_ballot_w64(m) ? b : 0.f; }
    } else if (MIX == 3) {   // 4 x (add chain -> cmp -> ballot -> scalar and -> select): VALU->SGPR->SALU->VALU round trip
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 += b;
        unsigned long long q = __ballot(a0 < a1);
        m &= q | 0x5555555555555555ull;
        a1 += __builtin_amdgcn_inverse_ballot_w64(m) ? b : 0.f;
      }
    } else if (MIX == 4) {   // exec windows: 8 x (s_mov exec + v_add) 4 chains
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        m = (m << 1) | (m >> 63);
        asm volatile("s_mov_b64 exec, %[m]\n\tv_add_f32 %[c0], %[c0], %[b]\n\ts_mov_b64 exec, -1" : [c0] "+v"(a0) : [m] "s"(m), [b] "v"(b));
        asm volatile("s_mov_b64 exec, %[m]\n\tv_add_f32 %[c0], %[c0], %[b]\n\ts_mov_b64 exec, -1" : [c0] "+v"(a1) : [m] "s"(m), [b] "v"(b));
        asm volatile("s_mov_b64 exec, %[m]\n\tv_add_f32 %[c0], %[c0], %[b]\n\ts_mov_b64 exec, -1" : [c0] "+v"(a2) : [m] "s"(m), [b] "v"(b));
        asm volatile("s_mov_b64 exec, %[m]\n\tv_add_f32 %[c0], %[c0], %[b]\n\ts_mov_b64 exec, -1" : [c0] "+v"(a3) : [m] "s"(m), [b] "v"(b));
      }
    } else if (MIX == 5) {   // 16 independent scalar ops
#pragma unroll
      for (int j = 0; j < 16; ++j) m = (m << 1) ^ (m >> 7);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + (float)m;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  const int iters = 20000;
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 64 * 64 * 4); hipMalloc(&cyc, 256 * 64 * 8);
  const char* names[6] = {"16 valu add, 4 chains", "16 valu add, 1 chain", "8x(s_shift, select, add) 1 chain", "4x(add, cmp->sgpr, s_and, select+add)", "8x(exec window add) 4 chains", "16 salu"};
  const int ninstr[6] = {16, 16, 32, 24, 26, 32};
  for (int mix = 0; mix < 6; ++mix)
    for (int wps : {1, 2, 4, 5, 6, 8}) {
      const int waves_per_cu = 4, grid = 256 * wps;            // wps workgroups of 4 waves per CU
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto launch = [&]() {
        switch (mix) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters, 12345ull); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters, 12345ull); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters, 12345ull); break;
          case 3: hipLaunchKernelGGL(k<3>, dim3(grid), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters, 12345ull); break;
          case 4: hipLaunchKernelGGL(k<4>, dim3(grid), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters, 12345ull); break;
          default: hipLaunchKernelGGL(k<5>, dim3(grid), dim3(64 * waves_per_cu), 0, 0, out, cyc, iters, 12345ull); break;
        }
      };
      launch(); hipDeviceSynchronize();
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double instr_per_simd = (double)iters * ninstr[mix] * wps;
      printf("%-40s waves/SIMD %d: %.3f ms  -> %.2f ns per instr per SIMD (%.2f instr/cycle/SIMD at 2.4 GHz), per wave %.1f cycles/instr\n", names[mix], wps, ms,
             ms * 1e6 / instr_per_simd, instr_per_simd / (ms * 1e-3 * 2.4e9), ms * 1e-3 * 2.4e9 / ((double)iters * ninstr[mix]));
    }
  return 0;
}
