// Deformable convolution forward for gfx950 (MI355X): the Lane-aware Feature
// Aggregator's gather-and-aggregate as ONE fused kernel - bilinear sampling of a
// channel-last feature map straight into an LDS tile, contracted on the matrix
// cores (exact-fp32 MFMA 32x32x2, or bf16 / f16 MFMA 32x32x16 with fp32 accumulate).
//
// Restates (arithmetic, not structure)
//   deformable_im2col_gpu_kernel / _bilinear
//       mmdet/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu:84-115, 190-243
//   modulated_deformable_im2col_gpu_kernel            same file :570-633
//   deform_conv_forward_cuda (GEMM orientation, output layout)
//       mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:152-260, :490-569
// The reference materialises columns[Cin*kH*kW, step*Hout*Wout] in HBM, re-reads
// it in a cuBLAS GEMM and transposes the result; here there is no column buffer:
// HBM sees x, the offsets and the output once.
//
// Work decomposition (M = B*Hout*Wout output pixels, N = Cout, K = taps*Cin):
//   workgroup = 256 threads = 4 wavefronts, tile = 128 pixels x 64 outputs,
//   wavefront w owns pixels [32w, 32w+32) x all 64 outputs (two 32x32 accumulators).
//   for tap, deformable group:   128 threads compute (corner index, 4 bilinear
//                                weights, validity bits) once per pixel -> LDS
//     for channel chunk (256 B of channels per pixel):
//        gather : 16 lanes x 16 B cover one pixel's chunk per corner -> every load
//                 instruction reads whole 256-B runs of the NHWC map (coalesced);
//                 blend 4 corners in fp32, write the A tile [128][chunk] to LDS
//        stage  : weight chunk [64][chunk] -> LDS (weights are pre-packed
//                 [tap][Cout][Cin] so rows are contiguous)
//        MFMA   : A and B fragments by ds_read_b128 (row stride 272 B: conflict
//                 free), k order permuted identically on both operands.
//   epilogue: optional bias / ReLU, NHWC direct from the accumulators or NCHW
//   through an LDS transpose so that stores stay coalesced.

#include "common.h"

namespace {

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

constexpr int TM = 128;       // output pixels per workgroup
constexpr int TN = 64;        // output channels per workgroup
constexpr int NTHREADS = 256;
constexpr int CHUNK_BYTES_MAX = 256;

struct DcnArgs {
  const void *xh;       // NHWC feature map, Cp channels per pixel
  const float *offset;  // [B][dg*2*T][Ho][Wo]
  const float *mask;    // [B][dg*T][Ho][Wo] or nullptr
  const void *wp;       // packed weights [T][CoutP][Cp]
  const void *bias;     // [Cout] or nullptr
  void *out;
  int B, H, W, Cp, Ho, Wo, Cout, CoutP;
  int kH, kW, sH, sW, pH, pW, dH, dW;
  int dg, cpg;          // deformable groups, channels per group (of Cp)
  int out_nhwc, relu, points;
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
  static constexpr int EPL = 4;
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
  static constexpr int EPL = 8;
  typedef bf16x8 frag;
  __device__ static __forceinline__ void mma(const frag &a, const frag &b,
                                             f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};

template <>
struct Mma<f16_t> {
  static constexpr int EPL = 8;
  typedef f16x8 frag;
  __device__ static __forceinline__ void mma(const frag &a, const frag &b,
                                             f32x16 &c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// 16-byte piece of EPL elements
template <typename TA>
struct alignas(16) Piece {
  TA e[16 / sizeof(TA)];
};

template <typename TA, typename TO, int KC>
__global__ __launch_bounds__(NTHREADS, 2) void dcn_fused_kernel(DcnArgs a) {
  constexpr int EPL = 16 / sizeof(TA);          // elements per 16-B piece
  constexpr int LPP = KC / EPL;                 // lanes (pieces) per pixel chunk
  constexpr int PPP = NTHREADS / LPP;           // pixels per gather pass
  constexpr int NPASS = TM / PPP;
  constexpr int LDA = KC + EPL;                 // padded row (elements): +16 B
  constexpr int GROUP = NPASS < 4 ? NPASS : 4;  // passes with loads in flight
  static_assert(KC * sizeof(TA) <= CHUNK_BYTES_MAX, "chunk too wide");
  static_assert(NPASS % GROUP == 0, "pass grouping");
  typedef Piece<TA> piece_t;
  typedef typename Mma<TA>::frag frag_t;

  constexpr int A_BYTES = TM * LDA * sizeof(TA);
  constexpr int B_BYTES = TN * LDA * sizeof(TA);
  constexpr int C_BYTES = TN * (TM + 1) * 4;    // NCHW epilogue transpose
  constexpr int AB_BYTES = (A_BYTES + B_BYTES) > C_BYTES ? (A_BYTES + B_BYTES) : C_BYTES;
  __shared__ __attribute__((aligned(16))) unsigned char smem[AB_BYTES];
  __shared__ int s_pix[TM];
  __shared__ float s_w[4][TM];
  __shared__ int s_flags[TM];
  __shared__ float s_mask[TM];

  TA *As = reinterpret_cast<TA *>(smem);
  TA *Bs = reinterpret_cast<TA *>(smem + A_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  const long long tile = blockIdx.x;
  const int n0 = blockIdx.y * TN;
  const int HWo = a.Ho * a.Wo;
  const int T = a.kH * a.kW;
  const TA *xh = static_cast<const TA *>(a.xh);
  const TA *wp = static_cast<const TA *>(a.wp);

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  const int lp = tid % LPP;  // piece within the pixel chunk
  const int pr = tid / LPP;  // pixel within the pass

  for (int t = 0; t < T; ++t) {
    const int ti = t / a.kW, tj = t - ti * a.kW;
    for (int g = 0; g < a.dg; ++g) {
      __syncthreads();  // previous readers of the coordinate table are done
      if (tid < TM) {
        const long long p = tile * TM + tid;
        int flags = 0, pix = 0;
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f, mk = 1.f;
        if (p < a.P) {
          const int b = (int)(p / HWo);
          const int hw = (int)(p - (long long)b * HWo);
          const int ho = hw / a.Wo, wo = hw - ho * a.Wo;
          const float *op = a.offset +
                            ((size_t)(b * a.dg + g) * 2 * T + 2 * t) * HWo + hw;
          float off_h = op[0];
          float off_w = op[HWo];
          if (a.points)  // dcn_offset = pts - base  (lanepoints_conv.py:63-74,170)
            off_w = off_w - (float)(tj - a.pW);
          const float h_im = (float)(ho * a.sH - a.pH + ti * a.dH) + off_h;
          const float w_im = (float)(wo * a.sW - a.pW + tj * a.dW) + off_w;
          if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw_ = 1.f - lw;
            w1 = hh * hw_;
            w2 = hh * lw;
            w3 = lh * hw_;
            w4 = lh * lw;
            if (h_low >= 0 && w_low >= 0) flags |= 1;
            if (h_low >= 0 && w_high <= a.W - 1) flags |= 2;
            if (h_high <= a.H - 1 && w_low >= 0) flags |= 4;
            if (h_high <= a.H - 1 && w_high <= a.W - 1) flags |= 8;
            pix = (b * a.H + h_low) * a.W + w_low;
          }
          if (a.mask)
            mk = a.mask[((size_t)(b * a.dg + g) * T + t) * HWo + hw];
        }
        s_pix[tid] = pix;
        s_w[0][tid] = w1;
        s_w[1][tid] = w2;
        s_w[2][tid] = w3;
        s_w[3][tid] = w4;
        s_flags[tid] = flags;
        s_mask[tid] = mk;
      }
      __syncthreads();

      const int c_begin = g * a.cpg, c_end = c_begin + a.cpg;
      for (int c0 = c_begin; c0 < c_end; c0 += KC) {
        // ---- stage the weight chunk [TN][KC] ------------------------------
        {
          constexpr int PIECES = TN * LPP;
#pragma unroll
          for (int q = tid; q < PIECES; q += NTHREADS) {
            const int row = q / LPP, pc = q - row * LPP;
            const piece_t v = *reinterpret_cast<const piece_t *>(
                wp + ((size_t)(t * a.CoutP + n0 + row) * a.Cp + c0 + pc * EPL));
            *reinterpret_cast<piece_t *>(Bs + row * LDA + pc * EPL) = v;
          }
        }
        // ---- gather + blend the A chunk [TM][KC] --------------------------
#pragma unroll
        for (int pg = 0; pg < NPASS; pg += GROUP) {
          piece_t v[GROUP][4];
          int fl[GROUP];
#pragma unroll
          for (int u = 0; u < GROUP; ++u) {
            const int r = (pg + u) * PPP + pr;
            const int f = s_flags[r];
            fl[u] = f;
            const TA *base = xh + (size_t)(long long)s_pix[r] * a.Cp + c0 + lp * EPL;
            const piece_t zero = {};
            v[u][0] = (f & 1) ? *reinterpret_cast<const piece_t *>(base) : zero;
            v[u][1] = (f & 2) ? *reinterpret_cast<const piece_t *>(base + a.Cp) : zero;
            v[u][2] = (f & 4) ? *reinterpret_cast<const piece_t *>(base + (size_t)a.W * a.Cp) : zero;
            v[u][3] = (f & 8) ? *reinterpret_cast<const piece_t *>(base + (size_t)(a.W + 1) * a.Cp) : zero;
          }
#pragma unroll
          for (int u = 0; u < GROUP; ++u) {
            const int r = (pg + u) * PPP + pr;
            const float w1 = s_w[0][r], w2 = s_w[1][r], w3 = s_w[2][r], w4 = s_w[3][r];
            const float mk = s_mask[r];
            piece_t o;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
              // w1*v1 + w2*v2 + w3*v3 + w4*v4   (deform_conv_cuda_kernel.cu:111-113)
              float val = w1 * to_f32(v[u][0].e[e]);
              val = fmaf(w2, to_f32(v[u][1].e[e]), val);
              val = fmaf(w3, to_f32(v[u][2].e[e]), val);
              val = fmaf(w4, to_f32(v[u][3].e[e]), val);
              if (a.mask) val = val * mk;  // :627
              o.e[e] = from_f32<TA>(val);
            }
            *reinterpret_cast<piece_t *>(As + r * LDA + lp * EPL) = o;
          }
        }
        __syncthreads();
        // ---- contract: wavefront `wid` x (2 x 32 outputs) -----------------
        {
          const int i = lane & 31, h = lane >> 5;
          const TA *arow = As + (wid * 32 + i) * LDA + h * EPL;
          const TA *brow0 = Bs + i * LDA + h * EPL;
          const TA *brow1 = Bs + (32 + i) * LDA + h * EPL;
#pragma unroll
          for (int q = 0; q < KC / (2 * EPL); ++q) {
            const frag_t fa = *reinterpret_cast<const frag_t *>(arow + q * 2 * EPL);
            const frag_t fb0 = *reinterpret_cast<const frag_t *>(brow0 + q * 2 * EPL);
            const frag_t fb1 = *reinterpret_cast<const frag_t *>(brow1 + q * 2 * EPL);
            Mma<TA>::mma(fa, fb0, acc0);
            Mma<TA>::mma(fa, fb1, acc1);
          }
        }
        __syncthreads();
      }
    }
  }

  // ---- epilogue -------------------------------------------------------------
  // accumulator map (32x32): column = lane & 31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  TO *out = static_cast<TO *>(a.out);
  const TO *bias = static_cast<const TO *>(a.bias);
  const int col = lane & 31, half = lane >> 5;
  float bias0 = 0.f, bias1 = 0.f;
  if (bias) {
    if (n0 + col < a.Cout) bias0 = to_f32(bias[n0 + col]);
    if (n0 + 32 + col < a.Cout) bias1 = to_f32(bias[n0 + 32 + col]);
  }
  if (a.out_nhwc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const long long p = tile * TM + wid * 32 + row;
      if (p < a.P) {
        float v0 = acc0[r] + bias0, v1 = acc1[r] + bias1;
        if (a.relu) {
          v0 = fmaxf(v0, 0.f);
          v1 = fmaxf(v1, 0.f);
        }
        if (n0 + col < a.Cout) out[p * a.Cout + n0 + col] = from_f32<TO>(v0);
        if (n0 + 32 + col < a.Cout) out[p * a.Cout + n0 + 32 + col] = from_f32<TO>(v1);
      }
    }
  } else {
    float *Cs = reinterpret_cast<float *>(smem);  // [TN][TM+1]; A/B are dead
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      Cs[col * (TM + 1) + row] = acc0[r] + bias0;
      Cs[(32 + col) * (TM + 1) + row] = acc1[r] + bias1;
    }
    __syncthreads();
    for (int idx = tid; idx < TN * TM; idx += NTHREADS) {
      const int o = idx / TM, r = idx - o * TM;
      const long long p = tile * TM + r;
      if (p < a.P && n0 + o < a.Cout) {
        const int b = (int)(p / HWo);
        const int hw = (int)(p - (long long)b * HWo);
        float v = Cs[o * (TM + 1) + r];
        if (a.relu) v = fmaxf(v, 0.f);
        out[((size_t)b * a.Cout + n0 + o) * HWo + hw] = from_f32<TO>(v);
      }
    }
  }
}

// ---- layout / packing helpers -----------------------------------------------

// x[B][C][HW] (TS) -> xh[B][HW][Cp] (TD), channels >= C zero-filled.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(
    const TS *__restrict__ src, TD *__restrict__ dst, int C, int HW, int Cp) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < 64; c += 4) {
    float v = 0.f;
    if (c0 + c < C && p0 + tx < HW)
      v = to_f32(src[((size_t)b * C + c0 + c) * HW + p0 + tx]);
    tile[c][tx] = v;
  }
  __syncthreads();
  for (int p = ty; p < 64; p += 4) {
    if (p0 + p < HW && c0 + tx < Cp)
      dst[((size_t)b * HW + p0 + p) * Cp + c0 + tx] = from_f32<TD>(tile[tx][p]);
  }
}

// xs[N][C] (TS) -> xh[N][Cp] (TD): channel padding and/or dtype change of an
// already channel-last map.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void nhwc_repack_kernel(
    const TS *__restrict__ src, TD *__restrict__ dst, long long n_pix, int C,
    int Cp) {
  const long long total = n_pix * Cp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const long long p = i / Cp;
    const int c = (int)(i - p * Cp);
    dst[i] = from_f32<TD>(c < C ? to_f32(src[p * C + c]) : 0.f);
  }
}

// W[Cout][Cin][T] (TS) -> Wp[T][CoutP][Cp] (TD), zero padded.
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void pack_weight_kernel(
    const TS *__restrict__ w, TD *__restrict__ wp, int Cout, int Cin, int T,
    int CoutP, int Cp) {
  const long long total = (long long)T * CoutP * Cp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int o = (int)((i / Cp) % CoutP);
    const int t = (int)(i / ((long long)Cp * CoutP));
    float v = 0.f;
    if (o < Cout && c < Cin) v = to_f32(w[((size_t)o * Cin + c) * T + t]);
    wp[i] = from_f32<TD>(v);
  }
}

// ---- host side ------------------------------------------------------------------

struct Plan {
  int Ho, Wo, T;
  int ta;        // arithmetic / LDS element type (ganet_dtype)
  int kc;        // channel chunk (elements)
  int Cp, CoutP;
  bool copy_x;   // a channel-last (and/or converted, padded) copy of x is made
  size_t xh_bytes, wp_bytes;
};

size_t dtype_size(int dt) { return dt == GANET_F32 ? 4 : 2; }

int make_plan(const ganet_dcn_params *p, Plan *pl) {
  if (!p) return GANET_ERR_NULL;
  if (p->B <= 0 || p->Cin <= 0 || p->H <= 0 || p->W <= 0 || p->Cout <= 0)
    return GANET_ERR_BAD_SHAPE;
  // shape_check, deform_conv_cuda.cpp:62-150
  if (p->kH <= 0 || p->kW <= 0 || p->sH <= 0 || p->sW <= 0 || p->dH <= 0 ||
      p->dW <= 0 || p->pH < 0 || p->pW < 0)
    return GANET_ERR_BAD_SHAPE;
  if (p->groups <= 0 || p->deformable_groups <= 0) return GANET_ERR_BAD_SHAPE;
  if (p->Cin % p->groups || p->Cout % p->groups ||
      p->Cin % p->deformable_groups)
    return GANET_ERR_BAD_SHAPE;
  if (p->dtype != GANET_F32 && p->dtype != GANET_BF16 && p->dtype != GANET_F16)
    return GANET_ERR_BAD_SHAPE;
  if ((p->x_layout != GANET_NCHW && p->x_layout != GANET_NHWC) ||
      (p->out_layout != GANET_NCHW && p->out_layout != GANET_NHWC))
    return GANET_ERR_BAD_SHAPE;
  pl->Ho = ganet_dcn_out_size(p->H, p->pH, p->dH, p->kH, p->sH);
  pl->Wo = ganet_dcn_out_size(p->W, p->pW, p->dW, p->kW, p->sW);
  if (pl->Ho < 1 || pl->Wo < 1) return GANET_ERR_TOO_SMALL;
  if (p->groups != 1) return GANET_ERR_UNSUPPORTED;
  if (p->offset_is_points &&
      (p->kH != 1 || p->dW != 1 || p->pW != (p->kW - 1) / 2 || !(p->kW & 1)))
    return GANET_ERR_BAD_SHAPE;
  if ((long long)p->B * p->H * p->W >= (1ll << 31) ||
      (long long)p->B * pl->Ho * pl->Wo >= (1ll << 31))
    return GANET_ERR_UNSUPPORTED;
  pl->T = p->kH * p->kW;
  pl->ta = (p->dtype == GANET_F32 && p->mfma_bf16) ? GANET_BF16 : p->dtype;
  const int kc_big = pl->ta == GANET_F32 ? 64 : 128;
  const int kc_small = kc_big / 2;
  const int cpg = p->Cin / p->deformable_groups;
  if (cpg % kc_big == 0) {
    pl->kc = kc_big;
    pl->Cp = p->Cin;
  } else if (cpg % kc_small == 0) {
    pl->kc = kc_small;
    pl->Cp = p->Cin;
  } else if (p->deformable_groups == 1) {
    pl->kc = kc_small;
    pl->Cp = (int)ganet_align_up((size_t)p->Cin, (size_t)kc_small);
  } else {
    return GANET_ERR_UNSUPPORTED;  // padded channels would straddle groups
  }
  pl->CoutP = (int)ganet_align_up((size_t)p->Cout, (size_t)TN);
  pl->copy_x = p->x_layout != GANET_NHWC || pl->Cp != p->Cin || pl->ta != p->dtype;
  pl->xh_bytes = pl->copy_x ? ganet_align_up((size_t)p->B * p->H * p->W * pl->Cp *
                                                 dtype_size(pl->ta), 256)
                            : 0;
  pl->wp_bytes = ganet_align_up(
      (size_t)pl->T * pl->CoutP * pl->Cp * dtype_size(pl->ta), 256);
  return GANET_OK;
}

template <typename TS, typename TD>
void launch_x_copy(const ganet_dcn_params *p, const Plan &pl, const void *x,
                   void *xh, hipStream_t s) {
  const int HW = p->H * p->W;
  if (p->x_layout == GANET_NCHW) {
    dim3 grid((HW + 63) / 64, (pl.Cp + 63) / 64, p->B);
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<TS, TD>), grid, dim3(256), 0, s,
                       static_cast<const TS *>(x), static_cast<TD *>(xh), p->Cin,
                       HW, pl.Cp);
  } else {
    const long long n_pix = (long long)p->B * HW;
    const long long total = n_pix * pl.Cp;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL((nhwc_repack_kernel<TS, TD>), dim3(blocks), dim3(256), 0, s,
                       static_cast<const TS *>(x), static_cast<TD *>(xh), n_pix,
                       p->Cin, pl.Cp);
  }
}

template <typename TS, typename TD>
void launch_pack_weight(const ganet_dcn_params *p, const Plan &pl, const void *w,
                        void *wp, hipStream_t s) {
  const long long total = (long long)pl.T * pl.CoutP * pl.Cp;
  const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
  hipLaunchKernelGGL((pack_weight_kernel<TS, TD>), dim3(blocks), dim3(256), 0, s,
                     static_cast<const TS *>(w), static_cast<TD *>(wp), p->Cout,
                     p->Cin, pl.T, pl.CoutP, pl.Cp);
}

template <typename TA, typename TO>
void launch_main(const Plan &pl, const DcnArgs &a, hipStream_t s) {
  const long long tiles = (a.P + TM - 1) / TM;
  dim3 grid((unsigned)tiles, (unsigned)(pl.CoutP / TN));
  constexpr int KC_BIG = sizeof(TA) == 4 ? 64 : 128;
  if (pl.kc == KC_BIG)
    hipLaunchKernelGGL((dcn_fused_kernel<TA, TO, KC_BIG>), grid, dim3(NTHREADS), 0, s, a);
  else
    hipLaunchKernelGGL((dcn_fused_kernel<TA, TO, KC_BIG / 2>), grid, dim3(NTHREADS), 0, s, a);
}

}  // namespace

extern "C" int ganet_dcn_out_size(int in, int pad, int dilation, int k,
                                  int stride) {
  if (stride <= 0) return 0;
  const int num = in + 2 * pad - (dilation * (k - 1) + 1);
  if (num < 0) return 0;  // Python floor division would give <= 0 here
  return num / stride + 1;
}

extern "C" size_t ganet_deform_conv_workspace_bytes(const ganet_dcn_params *p) {
  Plan pl;
  if (make_plan(p, &pl) != GANET_OK) return 0;
  return pl.xh_bytes + (p->weight_is_packed ? 0 : pl.wp_bytes);
}

extern "C" size_t ganet_deform_conv_packed_weight_bytes(const ganet_dcn_params *p) {
  Plan pl;
  if (make_plan(p, &pl) != GANET_OK) return 0;
  return pl.wp_bytes;
}

namespace {
int pack_weight_dispatch(const ganet_dcn_params *p, const Plan &pl,
                         const void *weight, void *wp, hipStream_t s) {
  if (p->dtype == GANET_F32 && pl.ta == GANET_F32)
    launch_pack_weight<float, float>(p, pl, weight, wp, s);
  else if (p->dtype == GANET_F32 && pl.ta == GANET_BF16)
    launch_pack_weight<float, bf16_t>(p, pl, weight, wp, s);
  else if (p->dtype == GANET_BF16)
    launch_pack_weight<bf16_t, bf16_t>(p, pl, weight, wp, s);
  else
    launch_pack_weight<f16_t, f16_t>(p, pl, weight, wp, s);
  return ganet_check_launch();
}
}  // namespace

extern "C" int ganet_deform_conv_pack_weight(const void *weight,
                                             const ganet_dcn_params *p,
                                             void *packed, size_t packed_bytes,
                                             ganet_stream_t stream) {
  if (!weight || !p || !packed) return GANET_ERR_NULL;
  Plan pl;
  const int rc = make_plan(p, &pl);
  if (rc != GANET_OK) return rc;
  if (packed_bytes < pl.wp_bytes) return GANET_ERR_WORKSPACE;
  return pack_weight_dispatch(p, pl, weight, packed, static_cast<hipStream_t>(stream));
}

extern "C" int ganet_deform_conv_forward(const void *x, const float *offset,
                                         const float *mask, const void *weight,
                                         const void *bias, void *out,
                                         const ganet_dcn_params *p,
                                         void *workspace, size_t workspace_bytes,
                                         ganet_stream_t stream) {
  if (!x || !offset || !weight || !out || !p) return GANET_ERR_NULL;
  Plan pl;
  const int rc = make_plan(p, &pl);
  if (rc != GANET_OK) return rc;
  // the reference refuses inputs smaller than the kernel for DCNv1
  // (deform_conv_cuda.cpp:127); the modules pad before calling (deform_conv.py:301-317)
  if (!mask && (p->H < p->kH || p->W < p->kW)) return GANET_ERR_BAD_SHAPE;
  if (bias && !mask) return GANET_ERR_BAD_SHAPE;  // DCNv1 has no bias (deform_conv.py:205)
  const size_t need = pl.xh_bytes + (p->weight_is_packed ? 0 : pl.wp_bytes);
  if (need > 0 && (workspace_bytes < need || !workspace))
    return GANET_ERR_WORKSPACE;

  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  void *xh = pl.copy_x ? static_cast<void *>(ws) : nullptr;
  const void *wp = p->weight_is_packed ? weight : static_cast<const void *>(ws + pl.xh_bytes);

  if (pl.copy_x) {
    if (p->dtype == GANET_F32 && pl.ta == GANET_F32)
      launch_x_copy<float, float>(p, pl, x, xh, s);
    else if (p->dtype == GANET_F32 && pl.ta == GANET_BF16)
      launch_x_copy<float, bf16_t>(p, pl, x, xh, s);
    else if (p->dtype == GANET_BF16)
      launch_x_copy<bf16_t, bf16_t>(p, pl, x, xh, s);
    else
      launch_x_copy<f16_t, f16_t>(p, pl, x, xh, s);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }
  if (!p->weight_is_packed &&
      pack_weight_dispatch(p, pl, weight, ws + pl.xh_bytes, s) != GANET_OK)
    return GANET_ERR_HIP;

  DcnArgs a;
  a.xh = pl.copy_x ? xh : x;
  a.offset = offset;
  a.mask = mask;
  a.wp = wp;
  a.bias = bias;
  a.out = out;
  a.B = p->B;
  a.H = p->H;
  a.W = p->W;
  a.Cp = pl.Cp;
  a.Ho = pl.Ho;
  a.Wo = pl.Wo;
  a.Cout = p->Cout;
  a.CoutP = pl.CoutP;
  a.kH = p->kH;
  a.kW = p->kW;
  a.sH = p->sH;
  a.sW = p->sW;
  a.pH = p->pH;
  a.pW = p->pW;
  a.dH = p->dH;
  a.dW = p->dW;
  a.dg = p->deformable_groups;
  a.cpg = pl.Cp / p->deformable_groups;
  a.out_nhwc = p->out_layout == GANET_NHWC;
  a.relu = p->fuse_relu;
  a.points = p->offset_is_points;
  a.P = (long long)p->B * pl.Ho * pl.Wo;

  if (pl.ta == GANET_F32)
    launch_main<float, float>(pl, a, s);
  else if (pl.ta == GANET_BF16 && p->dtype == GANET_F32)
    launch_main<bf16_t, float>(pl, a, s);
  else if (pl.ta == GANET_BF16)
    launch_main<bf16_t, bf16_t>(pl, a, s);
  else
    launch_main<f16_t, f16_t>(pl, a, s);
  return ganet_check_launch();
}
