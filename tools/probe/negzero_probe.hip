// What does ReLU (fmaxf(x, 0.f) -> v_max_f32) return for x = -0.0f on gfx950?  (relu_mask_words packs sign bits as "bit pattern != 0".)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* x, unsigned* out) { out[threadIdx.x] = __float_as_uint(fmaxf(x[threadIdx.x], 0.f)); }
int main() {
  float h[4] = {-0.0f, 0.0f, -1.0f, 1e-45f};
  float* d; unsigned* o; unsigned r[4];
  hipMalloc(&d, 16); hipMalloc(&o, 16);
  hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, d, o);
  hipMemcpy(r, o, 16, hipMemcpyDeviceToHost);
  printf("relu(-0)=%08x relu(+0)=%08x relu(-1)=%08x relu(denorm)=%08x\n", r[0], r[1], r[2], r[3]);
  return 0;
}
