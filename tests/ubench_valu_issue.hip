// micro-benchmarks: issue cost of VALU instructions for ONE wave per SIMD on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE> __global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  float a[16]; f2 b[16];
  for (int i = 0; i < 16; i++) { a[i] = out[threadIdx.x + i]; b[i] = f2{a[i], a[i] + 1.f}; }
  float s = out[threadIdx.x + 20], t = s + 0.5f;
  f2 sv = {s, t};
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; it++) {
    if (MODE == 0) {  // 16 independent v_fma_f32
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = __builtin_fmaf(a[i], s, t);
    } else if (MODE == 1) {  // dependent chain v_fma_f32
#pragma unroll
      for (int r = 0; r < 128; r++) a[0] = __builtin_fmaf(a[0], s, t);
    } else if (MODE == 2) {  // 16 independent v_pk_fma_f32
#pragma unroll
      for (int r = 0; r < 8; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) b[i] = b[i] * sv + sv;
    } else if (MODE == 3) {  // dependent chain v_pk_fma_f32
#pragma unroll
      for (int r = 0; r < 128; r++) b[0] = b[0] * sv + sv;
    } else if (MODE == 4) {  // two interleaved dependent chains (fma)
#pragma unroll
      for (int r = 0; r < 64; r++) { a[0] = __builtin_fmaf(a[0], s, t); a[1] = __builtin_fmaf(a[1], s, t); }
    } else if (MODE == 5) {  // 16 independent v_mul + dependent v_sub (mul+sub pattern)
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int i = 0; i < 16; i++) { float m = a[i] * s; a[i] = t - m; }
    }
  }
  long long t1 = clock64();
  float acc = 0; for (int i = 0; i < 16; i++) acc += a[i] + b[i].x + b[i].y;
  out[threadIdx.x] = acc;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int MODE> int run(const char* name, int threads, float* d, long long* c) {
  long long h[4];
  k<MODE><<<1, threads>>>(d, c, 1000);
  CHECK(hipDeviceSynchronize());
  k<MODE><<<1, threads>>>(d, c, 1000);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost));
  printf("%-40s threads %3d: %.2f cycles / instruction (wave 0)\n", name, threads, (double)h[0] / (1000.0 * 128));
  return 0;
}
int main() {
  float* d; long long* c;
  CHECK(hipMalloc(&d, 4096 * 4)); CHECK(hipMemset(d, 0, 4096 * 4)); CHECK(hipMalloc(&c, 64 * 8));
  for (int threads : {64, 256, 512, 1024}) {
    run<0>("16 independent v_fma_f32", threads, d, c);
    run<1>("dependent v_fma_f32 chain", threads, d, c);
    run<2>("16 independent v_pk_fma_f32", threads, d, c);
    run<3>("dependent v_pk_fma_f32 chain", threads, d, c);
    run<4>("2 interleaved dependent fma chains", threads, d, c);
    run<5>("mul + dependent sub x16", threads, d, c);
  }
  return 0;
}
