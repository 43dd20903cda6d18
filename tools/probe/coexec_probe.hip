// Does a VALU-heavy wave overlap with an MFMA-heavy wave on the same SIMD (gfx950)?  Workgroups of 512 threads: waves 0-3
// run role A, waves 4-7 role B (waves w and w + 4 share a SIMD).  Roles: 0 idle, 1 f16 MFMA loop (32x32x16, independent
// accumulators), 2 VALU loop (fma chain x 8 independent), 3 epilogue-like mix (cvt, pack, LDS 8-byte stores), 4 LDS tr reads + MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

__device__ __forceinline__ void role_mfma(int iters, float* out) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x ^ j)); }
  f32x16 c0 = {0}, c1 = {0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
    }
  }
  out[threadIdx.x] = c0[0] + c1[3];
}
__device__ __forceinline__ void role_valu(int iters, float* out) {
  float x[8];
  for (int j = 0; j < 8; ++j) x[j] = 0.5f + 0.001f * (threadIdx.x + j);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = __builtin_fmaf(x[j], 0.999f, 0.001f);
  }
  float s = 0; for (int j = 0; j < 8; ++j) s += x[j];
  out[threadIdx.x] = s;
}
__device__ __forceinline__ void role_epi(int iters, float* out, char* lds) {
  float x[16];
  for (int j = 0; j < 16; ++j) x[j] = 0.5f + 0.001f * (threadIdx.x + j);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      s16x4 v0, v1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y = fmaxf(x[4 * g + j] + 0.25f, 0.f) * 1.5f;
        const _Float16 t0 = (_Float16)y; const _Float16 t1 = (_Float16)(y - (float)t0);
        v0[j] = __builtin_bit_cast(short, t0); v1[j] = __builtin_bit_cast(short, t1);
        x[4 * g + j] = y * 0.37f;
      }
      *reinterpret_cast<s16x4*>(lds + ((threadIdx.x & 255) * 64 + g * 16)) = v0;
      *reinterpret_cast<s16x4*>(lds + ((threadIdx.x & 255) * 64 + g * 16 + 8)) = v1;
    }
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += x[j];
  out[threadIdx.x] = s;
}

__device__ __forceinline__ void role_mix(int iters, float* out, char* lds) {
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(0.001f * (threadIdx.x + j)); b[j] = (_Float16)(0.002f * (threadIdx.x ^ j)); }
  f32x16 c0 = {0}, c1 = {0};
  float x[16];
  for (int j = 0; j < 16; ++j) x[j] = 0.5f + 0.001f * (threadIdx.x + j);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c1, 0, 0, 0);
      s16x4 v0, v1;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float y = fmaxf(x[4 * g + j] + 0.25f, 0.f) * 1.5f;
        const _Float16 t0 = (_Float16)y; const _Float16 t1 = (_Float16)(y - (float)t0);
        v0[j] = __builtin_bit_cast(short, t0); v1[j] = __builtin_bit_cast(short, t1);
        x[4 * g + j] = y * 0.37f;
      }
      *reinterpret_cast<s16x4*>(lds + ((threadIdx.x & 255) * 64 + g * 16)) = v0;
      *reinterpret_cast<s16x4*>(lds + ((threadIdx.x & 255) * 64 + g * 16 + 8)) = v1;
    }
  }
  float s = 0; for (int j = 0; j < 16; ++j) s += x[j];
  out[threadIdx.x] = c0[0] + c1[3] + s;
}

__global__ __launch_bounds__(512, 2) void k(int roleA, int roleB, int itA, int itB, float* out, long long* cyc, int prioB) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int grp = threadIdx.x >> 8;
  const int role = grp ? roleB : roleA, it = grp ? itB : itA;
  if (__builtin_amdgcn_readfirstlane(grp) == 1) { if (prioB == 1) __builtin_amdgcn_s_setprio(1); else if (prioB == 2) __builtin_amdgcn_s_setprio(2); else if (prioB == 3) __builtin_amdgcn_s_setprio(3); }
  const long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 1) role_mfma(it, out + blockIdx.x * 512);
  else if (role == 2) role_valu(it, out + blockIdx.x * 512);
  else if (role == 3) role_epi(it, out + blockIdx.x * 512, lds + grp * 16384);
  else if (role == 5) role_mix(it, out + blockIdx.x * 512, lds + grp * 16384);
  const long long t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  float* out; long long* cyc;
  const int nb = 256;
  hipMalloc(&out, nb * 512 * 4); hipMalloc(&cyc, nb * 8 * 8);
  long long h[nb * 8];
  auto run = [&](int ra, int rb, int ia, int ib, const char* tag, int prioB = 0) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(nb), dim3(512), 32768, 0, ra, rb, ia, ib, out, cyc, prioB);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(nb), dim3(512), 32768, 0, ra, rb, ia, ib, out, cyc, prioB);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < nb; ++i) { for (int w = 0; w < 4; ++w) a += h[i * 8 + w]; for (int w = 4; w < 8; ++w) b += h[i * 8 + w]; }
    printf("%-28s wall %7.1f us   group A %9.0f cyc   group B %9.0f cyc\n", tag, ms * 1e3, a / (nb * 4), b / (nb * 4));
  };
  const int IM = 400, IV = 400, IE = 1500;
  run(1, 0, IM, 0, "mfma alone");
  run(2, 0, IV, 0, "valu alone");
  run(3, 0, IE, 0, "epilogue-mix alone");
  run(1, 1, IM, IM, "mfma + mfma");
  run(2, 2, IV, IV, "valu + valu");
  run(1, 2, IM, IV, "mfma + valu");
  run(1, 3, IM, IE, "mfma + epilogue-mix");
  run(3, 3, IE, IE, "epilogue-mix x2");
  run(5, 0, IM, 0, "ONE wave: 16 mfma + epilogue-mix per iter");
  run(1, 0, IM, 0, "mfma alone (16 per iter, same count)");
  run(3, 0, IM, 0, "epilogue-mix alone (same count)");
  run(5, 5, IM, IM, "mix x2 waves per SIMD");
  run(1, 2, IM, IV, "mfma + valu(prio1)", 1);
  run(1, 2, IM, IV, "mfma + valu(prio3)", 3);
  run(1, 3, IM, IE, "mfma + epi-mix(prio1)", 1);
  run(1, 3, IM, IE, "mfma + epi-mix(prio3)", 3);
  run(2, 1, IV, IM, "valu(A, older) + mfma(B)");
  run(3, 1, IE, IM, "epi-mix(A, older) + mfma(B)");
  return 0;
}
