// Device-side types and helpers shared by the gfx950 deform-conv kernels
// (csrc/deform_conv.hip: sample-then-contract, csrc/deform_conv_cts.hip:
// contract-then-sample).  Internal, not part of the public ABI.
#pragma once

#include <type_traits>

#include "common.h"

namespace ganet_dcn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

struct bf16_t {
  unsigned short v;
};
typedef _Float16 f16_t;

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) {
  return __uint_as_float(((unsigned int)v.v) << 16);
}
__device__ __forceinline__ float to_f32(f16_t v) { return (float)v; }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) {
  return v;
}
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) {
  // plain cast: round-to-nearest-even, NaN stays NaN (v_cvt_pk_bf16_f32)
  __bf16 b = (__bf16)v;
  bf16_t r;
  r.v = __builtin_bit_cast(unsigned short, b);
  return r;
}
template <>
__device__ __forceinline__ f16_t from_f32<f16_t>(float v) {
  return (f16_t)v;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define GANET_GLOBAL __attribute__((address_space(1)))

// Pointers that reach a kernel inside a by-value struct are generic ("flat") to
// the compiler; these casts give it global_load/global_store_dwordx4.
__device__ __forceinline__ u32x4 ldg128(const void *p) {
  return *(const GANET_GLOBAL u32x4 *)(p);
}
__device__ __forceinline__ float ldg_f32(const float *p) {
  return *(const GANET_GLOBAL float *)(p);
}

constexpr int TN = 64;        // output channels per workgroup
constexpr int TM_MAX = 256;   // largest pixel tile of a workgroup (8 waves x 32)
constexpr int CHUNK_BYTES_MAX = 256;
constexpr unsigned OOB_OFFSET = 0x80000000u;  // beyond any buffer window: reads 0

typedef float f32x2 __attribute__((ext_vector_type(2)));

struct DcnArgs {
  const void *xh;       // NHWC feature map, Cp channels per pixel
  const float *offset;  // [B][dg*2*T][Ho][Wo]
  const float *mask;    // [B][dg*T][Ho][Wo] or nullptr
  const void *wp;       // packed weights [T][CoutP][Cp]
  const void *wp_frag;  // their fragment-major copy (algorithm 5)
  const void *bias;     // [Cout] or nullptr
  void *out;
  int B, H, W, Cp, Ho, Wo, Cout, CoutP;
  int kH, kW, sH, sW, pH, pW, dH, dW;
  int dg, cpg;          // deformable groups, channels per group (of Cp)
  int out_nhwc, relu, points;
  int ablate;           // GANET_ABLATE builds only: bit0 skip MFMAs, bit1 skip gather loads
  long long P;          // B*Ho*Wo
};

// ---- MFMA wrappers: one 32x32 accumulator, K-slab of `EPL` elements per lane ---
template <typename TA>
struct Mma;

template <>
struct Mma<float> {
  // 16 B = 4 fp32 per lane per operand -> four 32x32x2 steps.
  // lane (i = l&31, h = l>>5) supplies A[i][k=h] and B[k=h][j=i]; with the lane
  // reading channels [8q+4h, 8q+4h+4) of its row, step m contracts channels
  // 8q+m (h=0) and 8q+4+m (h=1) on both operands alike.
  typedef f32x4 frag;
  __device__ static __forceinline__ void mma(const frag &a, const frag &b,
                                             f32x16 &c) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
      c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[m], c, 0, 0, 0);
  }
};

template <>
struct Mma<bf16_t> {
  // 16 B = 8 bf16 per lane: lane (r, h) holds A[r][8h+j], B[8h+j][r], j = 0..7.
  typedef bf16x8 frag;
  __device__ static __forceinline__ void mma(const frag &a, const frag &b,
                                             f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <>
struct Mma<f16_t> {
  typedef f16x8 frag;
  __device__ static __forceinline__ void mma(const frag &a, const frag &b,
                                             f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// contract-then-sample path (csrc/deform_conv_cts.hip)
bool cts_applicable(const ganet_dcn_params *p, int Cp, int CoutP);
// float32 tensors whose per-tap GEMM runs as three bf16 passes (weights packed as
// a high and a low bf16 plane)
bool cts_split(const ganet_dcn_params *p, int Cp);
// images per chunk (Y of a chunk stays inside the Infinity Cache); Y bytes of nb images
int cts_chunk_images(const ganet_dcn_params *p, int CoutP, int ta, bool split);
size_t cts_y_bytes(const ganet_dcn_params *p, int CoutP, int ta, int nb);
int cts_forward(const void *xh, const float *offset, const float *mask, const void *wq,
                const void *bias, void *out, const ganet_dcn_params *p, int ta, int kc,
                int Cp, int CoutP, int Ho, int Wo, void *ybuf, int nb, hipStream_t s);

// the same algorithm as one persistent launch (csrc/deform_conv_stream.hip): producer
// workgroups contract, consumer workgroups sample, Y travels through a cache-resident ring
bool stream_applicable(const ganet_dcn_params *p, int ta, int Cp, int CoutP, int Ho, int Wo);
size_t stream_workspace_bytes(const ganet_dcn_params *p, int CoutP);
int stream_forward(const void *xh, const float *offset, const float *mask, const void *wq,
                   const void *bias, void *out, const ganet_dcn_params *p, int ta, int Cp,
                   int CoutP, void *ws, hipStream_t s);

// the same algorithm with Y kept in the forming XCD's L2 (csrc/deform_conv_ring.hip): teams of
// workgroups on one XCD form Y tap by tap into a three-plane ring and sample it in the same launch
bool ring_applicable(const ganet_dcn_params *p, int ta, int Cp, int CoutP, int Ho, int Wo);
size_t ring_workspace_bytes(const ganet_dcn_params *p);
int ring_forward(const void *xh, const float *offset, const float *mask, const void *wq,
                 const void *bias, void *out, const ganet_dcn_params *p, int ta, int Cp,
                 void *ws, hipStream_t s);

// exact-float32 GEMM of the same file, for the backward (csrc/deform_conv_bwd.hip)
int cts_gemm_f32(const float *x, const float *wq, float *y, long long M, int K, int N,
                 hipStream_t s);
// the same product in three bf16 passes over high/low splits (float32-class accuracy);
// wq_split: two bf16 planes [N][K] (high parts, then low parts)
int cts_gemm_f32_split(const float *x, const void *wq_split, float *y, long long M, int K,
                       int N, hipStream_t s);

// y[M][ldy] (+)= x * wq^T with a long K and N <= 256 output columns (x read once)
int cts_gemm_f32_split_wide(const float *x, const void *wq_split, float *y, long long M, int K,
                            int N, int ldy, int accumulate, hipStream_t s);

}  // namespace ganet_dcn
