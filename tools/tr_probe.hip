// Probe: MFMA 32x32x16 bf16 with BOTH operands read from K-outer LDS images [k][col] through ds_read_b64_tr_b16.
//   C[m][n] = sum_k A[k][m] * B[k][n],  k = 0..31 (two MFMAs), m, n = 0..31
// build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LDB 192   // bytes per k row of a 64-column image (64*2 + 64)

__device__ inline bf16x8 frag(const unsigned char* img, int col0, int k0, int lane) {
  // lane (r = lane & 31, h = lane >> 5) needs k = k0 + 8h .. +7 of column col0 + r
  const int idx = lane & 15, q = idx >> 2, p = idx & 3;
  const unsigned char* a = img + (k0 + 8 * (lane >> 5) + q) * LDB + (col0 + 16 * ((lane >> 4) & 1) + 4 * p) * 2;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(a + 4 * LDB));
  union { s16x4 s[2]; bf16x8 b; } u;
  u.s[0] = lo; u.s[1] = hi;
  return u.b;
}

__global__ void probe(const float* A, const float* B, float* C) {   // A,B: [32 k][64 cols] fp32 (cols 0..31 used + offset test)
  __shared__ __attribute__((aligned(16))) unsigned char sa[32 * LDB], sb[32 * LDB];
  for (int i = threadIdx.x; i < 32 * 64; i += 64) {
    const int k = i / 64, c = i % 64;
    ((__bf16*)(sa + k * LDB))[c] = (__bf16)A[i];
    ((__bf16*)(sb + k * LDB))[c] = (__bf16)B[i];
  }
  __syncthreads();
  const int lane = threadIdx.x;
  f32x16 acc;
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  for (int kg = 0; kg < 2; ++kg) {
    const bf16x8 a = frag(sa, 32, 16 * kg, lane);   // columns 32..63 of A as the 32 rows m
    const bf16x8 b = frag(sb, 0, 16 * kg, lane);    // columns 0..31 of B
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), col = lane & 31;
    C[row * 32 + col] = acc[e];
  }
}

int main() {
  std::vector<float> A(32 * 64), B(32 * 64), C(32 * 32), R(32 * 32, 0.f);
  for (int i = 0; i < 32 * 64; ++i) { A[i] = (float)((i * 7) % 13 - 6); B[i] = (float)((i * 5) % 11 - 5); }
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += A[k * 64 + 32 + m] * B[k * 64 + n]; R[m * 32 + n] = s; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, C.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 32 * 32; ++i) err = fmax(err, fabs(C[i] - R[i]));
  printf("tr_probe max abs err %g (C[0]=%g ref %g, C[33]=%g ref %g)\n", err, C[0], R[0], C[33], R[33]);
  return err == 0.0 ? 0 : 1;
}
