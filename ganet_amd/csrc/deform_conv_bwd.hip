// Deformable convolution backward for gfx950 (MI355X), float32 - the training side of
// the drop-in.
//
// Replaces deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda
// (mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:262-488) and
// modulated_deform_conv_cuda_backward (:571-700); kernel arithmetic:
// deformable_col2im (deform_conv_cuda_kernel.cu:279-330), deformable_col2im_coord
// (:363-427), get_gradient_weight / get_coordinate_weight (:117-188) and the modulated
// forms (:636-755).
//
// The reference materialises grad_columns = W^T * grad_out ([Cin*T][pixels], ~1 GB at
// the LFA's training shapes) and then runs col2im / col2im_coord over it.  Here the
// columns never exist in memory:
//
//   dcn_bwd_input_kernel   a wavefront owns 32 output pixels.  Per (tap, 32-channel
//       tile) it forms grad_col[32 px][32 c] on the matrix cores (fp32 32x32x2, K =
//       Cout) and consumes the accumulators in place: 4 corner atomics into grad_x
//       (channel-last, so one accumulator register is two 128-byte runs - the shape
//       float atomics run at full rate on), 4 corner reads of x for the coordinate
//       and mask gradients, which are reduced over channels in registers and over
//       lanes by shuffles.
//   dcn_bwd_weight_kernel  a workgroup owns one (tap, 32-channel tile) of grad_weight
//       and a share of the pixels: it re-samples the columns (bilinear blend in
//       registers) and contracts them with grad_out on the matrix cores
//       (D[o][c] += go[p][o] * col[p][c], K = pixels), flushing once with atomics.
//   dcn_bwd_bias_kernel    column sums of grad_out (DCNv2).
//
// Sums are fp32 and atomic, so results are order dependent at the last bits - as the
// reference's own atomicAdd col2im is.

#include "dcn_device.h"

namespace ganet_dcn {

namespace {

struct BwdArgs {
  const float *xh;      // [B*H*W][Cp] channel-last input
  const float *offset;  // [B][dg*2T][Ho][Wo]
  const float *mask;    // [B][dg*T][Ho][Wo] or nullptr
  const float *wp;      // [T][Cout][Cp] packed weights
  const float *go;      // [B*Ho*Wo][Cout] channel-last grad_out
  float *gxh;           // [B*H*W][Cp], accumulated into
  float *goff;          // like offset, zeroed before the launch, accumulated into
  float *gmask;         // like mask or nullptr, zeroed, accumulated into
  float *gwp;           // [T][Cout][Cp], zeroed, accumulated into
  int B, H, W, Cp, Ho, Wo, Cout;
  int kH, kW, sH, sW, pH, pW, dH, dW;
  int dg, cpg;
  long long P;          // B*Ho*Wo
};

// one sampling position: rows of xh of the four corners (-1: outside the map) and the
// bilinear geometry, deform_conv_cuda_kernel.cu:84-115
struct Sample {
  int c1, c2, c3, c4;
  float lh, lw, m;
};

__device__ __forceinline__ Sample make_sample(const BwdArgs &a, long long p, int t, int g) {
  Sample s;
  s.c1 = s.c2 = s.c3 = s.c4 = -1;
  s.lh = s.lw = 0.f;
  s.m = 0.f;
  if (p >= a.P) return s;
  const int HWo = a.Ho * a.Wo, T = a.kH * a.kW;
  const int b = (int)(p / HWo);
  const int hw = (int)(p - (long long)b * HWo);
  const int ho = hw / a.Wo, wo = hw - ho * a.Wo;
  const int ti = t / a.kW, tj = t - ti * a.kW;
  const float *op = a.offset + ((size_t)(b * a.dg + g) * 2 * T + 2 * t) * HWo + hw;
  const float h_im = (float)(ho * a.sH - a.pH + ti * a.dH) + ldg_f32(op);
  const float w_im = (float)(wo * a.sW - a.pW + tj * a.dW) + ldg_f32(op + HWo);
  s.m = a.mask ? ldg_f32(a.mask + ((size_t)(b * a.dg + g) * T + t) * HWo + hw) : 1.f;
  if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    s.lh = h_im - (float)h_low;
    s.lw = w_im - (float)w_low;
    const int base = (b * a.H + h_low) * a.W + w_low;
    const bool hl = h_low >= 0, hh = h_high <= a.H - 1;
    const bool wl = w_low >= 0, wh = w_high <= a.W - 1;
    if (hl && wl) s.c1 = base;
    if (hl && wh) s.c2 = base + 1;
    if (hh && wl) s.c3 = base + a.W;
    if (hh && wh) s.c4 = base + a.W + 1;
  }
  return s;
}

constexpr int TAB = 8;  // words per table entry

__device__ __forceinline__ void put_sample(int *tab, int px, const Sample &s) {
  int *e = tab + px * TAB;
  e[0] = s.c1;
  e[1] = s.c2;
  e[2] = s.c3;
  e[3] = s.c4;
  e[4] = __float_as_int(s.lh);
  e[5] = __float_as_int(s.lw);
  e[6] = __float_as_int(s.m);
  e[7] = 0;
}
__device__ __forceinline__ Sample get_sample(const int *tab, int px) {
  const int4 u = *reinterpret_cast<const int4 *>(tab + px * TAB);
  const int4 v = *reinterpret_cast<const int4 *>(tab + px * TAB + 4);
  Sample s;
  s.c1 = u.x;
  s.c2 = u.y;
  s.c3 = u.z;
  s.c4 = u.w;
  s.lh = __int_as_float(v.x);
  s.lw = __int_as_float(v.y);
  s.m = __int_as_float(v.z);
  return s;
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ float half_wave_sum(float v) {
  // over the 32 lanes that share lane>>5
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

constexpr int BW_WAVES = 4;

// ---- grad_x, grad_offset, grad_mask ---------------------------------------------
__global__ __launch_bounds__(BW_WAVES * 64) void dcn_bwd_input_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) int s_tab[BW_WAVES][32 * TAB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long tile = (long long)blockIdx.x * BW_WAVES + wid;
  const long long p0 = tile * 32;
  if (p0 >= a.P) return;   // whole wave out (no workgroup barrier in this kernel)
  int *tab = s_tab[wid];
  const int T = a.kH * a.kW, HWo = a.Ho * a.Wo;
  const long long pa = p0 + i;      // the pixel this lane feeds into the A operand

  for (int ob = 0; ob < a.Cout; ob += 64) {
    // A operand: grad_out[p][o], lane (i = pixel, h): o = ob + 2m + h
    float ga[32];
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const int o = ob + 2 * m + h;
      ga[m] = (pa < a.P && o < a.Cout) ? ldg_f32(a.go + pa * a.Cout + o) : 0.f;
    }
    for (int g = 0; g < a.dg; ++g) {
      for (int t = 0; t < T; ++t) {
        if (lane < 32) put_sample(tab, i, make_sample(a, p0 + i, t, g));
        wave_sync_lds();
        float dh[16], dw[16], dm[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r] = dw[r] = dm[r] = 0.f;
        const int ct0 = g * a.cpg / 32, ct1 = (g + 1) * a.cpg / 32;
        for (int ct = ct0; ct < ct1; ++ct) {
          const int c = ct * 32 + i;
          f32x16 acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
          // grad_col[px][c] = sum_o go[px][o] * W[o][c][t]
          const float *wrow = a.wp + ((size_t)t * a.Cout + ob + h) * a.Cp + c;
#pragma unroll
          for (int m = 0; m < 32; ++m) {
            const float bv = (ob + 2 * m + h < a.Cout) ? ldg_f32(wrow + (size_t)2 * m * a.Cp) : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ga[m], bv, acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
            const Sample s = get_sample(tab, px);
            const float gc = acc[r];
            const float gm = gc * s.m;
            const float hh = 1.f - s.lh, hw = 1.f - s.lw;
            float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
            if (s.c1 >= 0) {
              v1 = ldg_f32(a.xh + (size_t)s.c1 * a.Cp + c);
              unsafeAtomicAdd(a.gxh + (size_t)s.c1 * a.Cp + c, hh * hw * gm);
            }
            if (s.c2 >= 0) {
              v2 = ldg_f32(a.xh + (size_t)s.c2 * a.Cp + c);
              unsafeAtomicAdd(a.gxh + (size_t)s.c2 * a.Cp + c, hh * s.lw * gm);
            }
            if (s.c3 >= 0) {
              v3 = ldg_f32(a.xh + (size_t)s.c3 * a.Cp + c);
              unsafeAtomicAdd(a.gxh + (size_t)s.c3 * a.Cp + c, s.lh * hw * gm);
            }
            if (s.c4 >= 0) {
              v4 = ldg_f32(a.xh + (size_t)s.c4 * a.Cp + c);
              unsafeAtomicAdd(a.gxh + (size_t)s.c4 * a.Cp + c, s.lh * s.lw * gm);
            }
            // get_coordinate_weight, deform_conv_cuda_kernel.cu:146-188
            dh[r] += gm * (-hw * v1 - s.lw * v2 + hw * v3 + s.lw * v4);
            dw[r] += gm * (-hh * v1 + hh * v2 - s.lh * v3 + s.lh * v4);
            dm[r] += gc * (hh * hw * v1 + hh * s.lw * v2 + s.lh * hw * v3 + s.lh * s.lw * v4);
          }
        }
        // channels of this group -> one value per (pixel, tap, direction)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sh = half_wave_sum(dh[r]);
          const float sw = half_wave_sum(dw[r]);
          const float sm = a.gmask ? half_wave_sum(dm[r]) : 0.f;
          const int px = (r & 3) + 8 * (r >> 2) + 4 * h;
          const long long p = p0 + px;
          if (i == 0 && p < a.P) {
            const int b = (int)(p / HWo);
            const int hw_ = (int)(p - (long long)b * HWo);
            float *q = a.goff + ((size_t)(b * a.dg + g) * 2 * T + 2 * t) * HWo + hw_;
            unsafeAtomicAdd(q, sh);
            unsafeAtomicAdd(q + HWo, sw);
            if (a.gmask)
              unsafeAtomicAdd(a.gmask + ((size_t)(b * a.dg + g) * T + t) * HWo + hw_, sm);
          }
        }
        wave_sync_lds();  // the table is rewritten for the next tap
      }
    }
  }
}

// ---- grad_weight -------------------------------------------------------------------
// grid.x = T * (Cp / 32): one (tap, channel tile); grid.y: shares of the pixel tiles
__global__ __launch_bounds__(BW_WAVES * 64) void dcn_bwd_weight_kernel(BwdArgs a) {
  __shared__ __attribute__((aligned(16))) int s_tab[BW_WAVES][32 * TAB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int nct = a.Cp / 32;
  const int t = blockIdx.x / nct, ct = blockIdx.x - t * nct;
  const int g = (ct * 32) / a.cpg;
  const int c = ct * 32 + i;
  int *tab = s_tab[wid];
  const long long ntiles = (a.P + 31) / 32;

  for (int ob = 0; ob < a.Cout; ob += 64) {
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    for (long long tile = (long long)blockIdx.y * BW_WAVES + wid; tile < ntiles;
         tile += (long long)gridDim.y * BW_WAVES) {
      const long long p0 = tile * 32;
      if (lane < 32) put_sample(tab, i, make_sample(a, p0 + i, t, g));
      wave_sync_lds();
#pragma unroll 4
      for (int m = 0; m < 16; ++m) {
        const int px = 2 * m + h;
        const long long p = p0 + px;
        const Sample s = get_sample(tab, px);
        const float hh = 1.f - s.lh, hw = 1.f - s.lw;
        float col = 0.f;
        if (s.c1 >= 0) col += hh * hw * ldg_f32(a.xh + (size_t)s.c1 * a.Cp + c);
        if (s.c2 >= 0) col += hh * s.lw * ldg_f32(a.xh + (size_t)s.c2 * a.Cp + c);
        if (s.c3 >= 0) col += s.lh * hw * ldg_f32(a.xh + (size_t)s.c3 * a.Cp + c);
        if (s.c4 >= 0) col += s.lh * s.lw * ldg_f32(a.xh + (size_t)s.c4 * a.Cp + c);
        col *= s.m;
        float a0 = 0.f, a1 = 0.f;
        if (p < a.P) {
          if (ob + i < a.Cout) a0 = ldg_f32(a.go + p * a.Cout + ob + i);
          if (ob + 32 + i < a.Cout) a1 = ldg_f32(a.go + p * a.Cout + ob + 32 + i);
        }
        // D[o][c] += go[p][o] * col[p][c], k = the pixel pair (2m, 2m+1)
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, col, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, col, acc1, 0, 0, 0);
      }
      wave_sync_lds();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int o = ob + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (o < a.Cout) unsafeAtomicAdd(a.gwp + ((size_t)t * a.Cout + o) * a.Cp + c, acc0[r]);
      if (o + 32 < a.Cout)
        unsafeAtomicAdd(a.gwp + ((size_t)t * a.Cout + o + 32) * a.Cp + c, acc1[r]);
    }
  }
}

// grad_bias[o] += sum_p go[p][o]
__global__ __launch_bounds__(256) void dcn_bwd_bias_kernel(const float *__restrict__ go,
                                                           float *__restrict__ gb,
                                                           long long P, int Cout) {
  const int o0 = threadIdx.x & 63, sub = threadIdx.x >> 6;
  for (int o = o0; o < Cout; o += 64) {
    float s = 0.f;
    for (long long p = (long long)blockIdx.x * 4 + sub; p < P; p += (long long)gridDim.x * 4)
      s += go[p * Cout + o];
    unsafeAtomicAdd(gb + o, s);
  }
}

// ---- layout helpers ------------------------------------------------------------------
// src[B][C][HW] -> dst[B*HW][Cp] (zero padded channels)
template <typename TS>
__global__ __launch_bounds__(256) void bwd_nchw_to_nhwc_kernel(const TS *__restrict__ src,
                                                               float *__restrict__ dst,
                                                               int C, int HW, int Cp) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int cc = ty; cc < 64; cc += 4) {
    float v = 0.f;
    if (c0 + cc < C && p0 + tx < HW) v = to_f32(src[((size_t)b * C + c0 + cc) * HW + p0 + tx]);
    tile[cc][tx] = v;
  }
  __syncthreads();
  for (int pp = ty; pp < 64; pp += 4)
    if (p0 + pp < HW && c0 + tx < Cp) dst[((size_t)b * HW + p0 + pp) * Cp + c0 + tx] = tile[tx][pp];
}

// dst[B][C][HW] += src[B*HW][Cp]
template <typename TD>
__global__ __launch_bounds__(256) void bwd_add_nhwc_to_nchw_kernel(const float *__restrict__ src,
                                                                   TD *__restrict__ dst,
                                                                   int C, int HW, int Cp) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const int c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int pp = ty; pp < 64; pp += 4) {
    float v = 0.f;
    if (p0 + pp < HW && c0 + tx < Cp) v = src[((size_t)b * HW + p0 + pp) * Cp + c0 + tx];
    tile[pp][tx] = v;
  }
  __syncthreads();
  for (int cc = ty; cc < 64; cc += 4)
    if (c0 + cc < C && p0 + tx < HW) {
      TD *d = dst + ((size_t)b * C + c0 + cc) * HW + p0 + tx;
      *d = from_f32<TD>(to_f32(*d) + tile[tx][cc]);
    }
}

// src[N][C] -> dst[N][Cp] (zero padded); and dst[N][C] += src[N][Cp]
template <typename TS>
__global__ __launch_bounds__(256) void bwd_pad_rows_kernel(const TS *__restrict__ src,
                                                           float *__restrict__ dst, long long n,
                                                           int C, int Cp) {
  const long long total = n * Cp;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const long long r = k / Cp;
    const int cc = (int)(k - r * Cp);
    dst[k] = cc < C ? to_f32(src[r * C + cc]) : 0.f;
  }
}
template <typename TD>
__global__ __launch_bounds__(256) void bwd_add_rows_kernel(const float *__restrict__ src,
                                                           TD *__restrict__ dst, long long n,
                                                           int C, int Cp) {
  const long long total = n * C;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const long long r = k / C;
    const int cc = (int)(k - r * C);
    dst[k] = from_f32<TD>(to_f32(dst[k]) + src[r * Cp + cc]);
  }
}

// W[Cout][Cin/groups][T] -> wp[T][Cout][Cp]; and gw[Cout][Cin/groups][T] += gwp[T][Cout][Cp].
// Convolution groups (deform_conv_cuda.cpp:227-236, :439-470) are a block-diagonal dense
// weight here, as in the forward: entries outside a row's group are packed as zeros, and
// the dense weight gradient's entries outside the groups are simply not read back.
template <typename TS>
__global__ __launch_bounds__(256) void bwd_pack_weight_kernel(const TS *__restrict__ w,
                                                              float *__restrict__ wp, int Cout,
                                                              int Cin, int T, int Cp, int groups) {
  const int cig = Cin / groups, cog = Cout / groups;
  const long long total = (long long)T * Cout * Cp;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const int cc = (int)(k % Cp);
    const int o = (int)((k / Cp) % Cout);
    const int t = (int)(k / ((long long)Cp * Cout));
    const int cg = cc - (o / cog) * cig;
    wp[k] = (cc < Cin && cg >= 0 && cg < cig) ? to_f32(w[((size_t)o * cig + cg) * T + t]) : 0.f;
  }
}
template <typename TD>
__global__ __launch_bounds__(256) void bwd_unpack_add_weight_kernel(
    const float *__restrict__ gwp, TD *__restrict__ gw, int Cout, int Cin, int T, int Cp,
    int groups) {
  const int cig = Cin / groups, cog = Cout / groups;
  const long long total = (long long)Cout * cig * T;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const int t = (int)(k % T);
    const int cg = (int)((k / T) % cig);
    const int o = (int)(k / ((long long)T * cig));
    const int cc = (o / cog) * cig + cg;
    gw[k] = from_f32<TD>(to_f32(gw[k]) + gwp[((size_t)t * Cout + o) * Cp + cc]);
  }
}

// gw[Cout][Cin/groups][T] += sum over planes of part[plane][T*Cout][Cp], planes added in
// order.  Threads walk the planes' own layout, so the reads are coalesced; the one write per
// element is the scattered side.
template <typename TD>
__global__ __launch_bounds__(256) void bwd_reduce_unpack_weight_kernel(
    const float *__restrict__ part, int planes, TD *__restrict__ gw, int Cout, int Cin, int T,
    int Cp, int groups) {
  const int cig = Cin / groups, cog = Cout / groups;
  const long long total = (long long)T * Cout * Cp;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const int cc = (int)(k % Cp);
    const int n = (int)(k / Cp);
    const int o = n % Cout, t = n / Cout;
    const int cg = cc - (o / cog) * cig;
    if (cc >= Cin || cg < 0 || cg >= cig) continue;
    float sum = 0.f;
    int pn = 0;
    for (; pn + 8 <= planes; pn += 8) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(pn + j) * total + k];
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    for (; pn < planes; ++pn) sum += part[(size_t)pn * total + k];
    const size_t dst = ((size_t)o * cig + cg) * T + t;
    gw[dst] = from_f32<TD>(to_f32(gw[dst]) + sum);
  }
}

// ===== commuted form: scatter / gather Cout-wide rows of the per-tap products ===========
// With Y_t = X * W_t^T (the forward's contract-then-sample intermediate) the backward is
//   grad_Y_t[q][o] += bilinear weight * mask * grad_out[p][o]        (scatter, atomics)
//   grad_x          = grad_Y * Wq                                     (one GEMM)
//   grad_Wq[n][c]   = sum_q grad_Y[q][n] * x[q][c]                    (one GEMM over pixels)
//   grad_offset, grad_mask from sum_o grad_out[p][o] * Y_t[corner][o]
// since sum_c grad_col[c] * x[corner][c] = sum_o grad_out[o] * Y_t[corner][o].  The rows
// that move through atomics are Cout wide instead of Cin wide (4x fewer bytes at 256 -> 64),
// which is what the direct kernels above are bound by.  Same arithmetic per term as
// deformable_col2im / col2im_coord (deform_conv_cuda_kernel.cu:279-427).
struct BwdCtsArgs {
  const float *y;        // [nimg*H*W][N] per-tap products of this chunk (nullptr: no coordinate gradients)
  float *gy;             // same shape, zeroed, accumulated into (nullptr: cts_bwd_bin_kernel forms it)
  const float *go;       // [B*Ho*Wo][Cout] channel-last grad_out (whole batch)
  const float *offset;   // whole batch
  const float *mask;     // whole batch or nullptr
  float *goff;           // zeroed, accumulated into
  float *gmask;          // zeroed, accumulated into, or nullptr
  unsigned *geo_q;       // binned form: [nimg][T][Ho*Wo] packed (h_low + 1) << 16 | (w_low + 1), ~0u = outside
  f32x2 *geo_f;          // same shape: (lh, lw)
  int img0, nimg;
  int H, W, Ho, Wo, Cout, T;
  int kW, pH, pW, dH, dW;
};

constexpr int CS_WAVES = 4;
constexpr int CS_TAPS = 32;   // taps whose geometry sits in LDS at a time

// 32 lanes per pixel, lane l owns channels l and l + 32 of one 64-channel block
// (blockIdx.y): every atomic instruction then adds one whole 128-byte run per pixel - the
// shape float atomics run at full rate on (16 lanes x 4 channels, 16-byte loads and four
// scattered words per lane, ran the scatter 3x slower).  2 pixels per wavefront.
__global__ __launch_bounds__(CS_WAVES * 64) void cts_bwd_sample_kernel(BwdCtsArgs a) {
  __shared__ __attribute__((aligned(16))) int s_geo[CS_WAVES][2][CS_TAPS][8];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pj = lane >> 5, l = lane & 31;
  const int cb = blockIdx.y;
  const int HWo = a.Ho * a.Wo;
  const int N = a.T * a.Cout;
  const long long P = (long long)a.nimg * HWo;
  const long long ngroups = (P + 1) / 2;
  const int c0 = cb * 64 + l;          // and c0 + 32; Cout is a multiple of 64 on this path

  for (long long grp = (long long)blockIdx.x * CS_WAVES + wid; grp < ngroups;
       grp += (long long)gridDim.x * CS_WAVES) {
    const long long p = grp * 2 + pj;
    const bool valid = p < P;
    const int bl = valid ? (int)(p / HWo) : 0;
    const int hw = valid ? (int)(p - (long long)bl * HWo) : 0;
    const int ho = hw / a.Wo, wo = hw - ho * a.Wo;
    const int b = a.img0 + bl;
    float g0 = 0.f, g1 = 0.f;
    if (valid) {
      const float *gp = a.go + ((size_t)b * HWo + hw) * a.Cout + c0;
      g0 = ldg_f32(gp);
      g1 = ldg_f32(gp + 32);
    }

    for (int t0 = 0; t0 < a.T; t0 += CS_TAPS) {
      const int nt = min(CS_TAPS, a.T - t0);
      if (l < nt) {
        const int t = t0 + l;
        const int ti = t / a.kW, tj = t - ti * a.kW;
        int c1 = -1, c2 = -1, c3 = -1, c4 = -1;
        float lh = 0.f, lw = 0.f, m = 0.f;
        if (valid) {
          const float *op = a.offset + ((size_t)b * 2 * a.T + 2 * t) * HWo + hw;
          const float h_im = (float)(ho - a.pH + ti * a.dH) + ldg_f32(op);
          const float w_im = (float)(wo - a.pW + tj * a.dW) + ldg_f32(op + HWo);
          m = a.mask ? ldg_f32(a.mask + ((size_t)b * a.T + t) * HWo + hw) : 1.f;
          if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            lh = h_im - (float)h_low;
            lw = w_im - (float)w_low;
            // float index of the corner's row of tap t inside the chunk's Y
            const int base = ((bl * a.H + h_low) * a.W + w_low) * N + t * a.Cout;
            const bool hl = h_low >= 0, hh = h_high <= a.H - 1;
            const bool wl = w_low >= 0, wh = w_high <= a.W - 1;
            if (hl && wl) c1 = base;
            if (hl && wh) c2 = base + N;
            if (hh && wl) c3 = base + a.W * N;
            if (hh && wh) c4 = base + a.W * N + N;
          }
        }
        int *e = s_geo[wid][pj][l];
        *reinterpret_cast<int4 *>(e) = make_int4(c1, c2, c3, c4);
        *reinterpret_cast<int4 *>(e + 4) =
            make_int4(__float_as_int(lh), __float_as_int(lw), __float_as_int(m), 0);
      }
      wave_sync_lds();
#pragma unroll 2
      for (int tt = 0; tt < nt; ++tt) {
        const int4 cx = *reinterpret_cast<const int4 *>(s_geo[wid][pj][tt]);
        const int4 ge = *reinterpret_cast<const int4 *>(s_geo[wid][pj][tt] + 4);
        const float lh = __int_as_float(ge.x), lw = __int_as_float(ge.y), m = __int_as_float(ge.z);
        const float hh = 1.f - lh, hw_ = 1.f - lw;
        const int t = t0 + tt;
        if (a.y) {
          float v1a = 0.f, v1b = 0.f, v2a = 0.f, v2b = 0.f, v3a = 0.f, v3b = 0.f, v4a = 0.f, v4b = 0.f;
          if (cx.x >= 0) { v1a = ldg_f32(a.y + cx.x + c0); v1b = ldg_f32(a.y + cx.x + c0 + 32); }
          if (cx.y >= 0) { v2a = ldg_f32(a.y + cx.y + c0); v2b = ldg_f32(a.y + cx.y + c0 + 32); }
          if (cx.z >= 0) { v3a = ldg_f32(a.y + cx.z + c0); v3b = ldg_f32(a.y + cx.z + c0 + 32); }
          if (cx.w >= 0) { v4a = ldg_f32(a.y + cx.w + c0); v4b = ldg_f32(a.y + cx.w + c0 + 32); }
          // get_coordinate_weight, deform_conv_cuda_kernel.cu:146-188
          float sh = g0 * (-hw_ * v1a - lw * v2a + hw_ * v3a + lw * v4a) +
                     g1 * (-hw_ * v1b - lw * v2b + hw_ * v3b + lw * v4b);
          float sw = g0 * (-hh * v1a + hh * v2a - lh * v3a + lh * v4a) +
                     g1 * (-hh * v1b + hh * v2b - lh * v3b + lh * v4b);
          float sm = g0 * (hh * hw_ * v1a + hh * lw * v2a + lh * hw_ * v3a + lh * lw * v4a) +
                     g1 * (hh * hw_ * v1b + hh * lw * v2b + lh * hw_ * v3b + lh * lw * v4b);
          sh = half_wave_sum(sh);
          sw = half_wave_sum(sw);
          if (a.gmask) sm = half_wave_sum(sm);
          if (l == 0 && valid) {
            float *q = a.goff + ((size_t)b * 2 * a.T + 2 * t) * HWo + hw;
            unsafeAtomicAdd(q, m * sh);
            unsafeAtomicAdd(q + HWo, m * sw);
            if (a.gmask) unsafeAtomicAdd(a.gmask + ((size_t)b * a.T + t) * HWo + hw, sm);
          }
        }
        if (valid && a.gy) {
          const float w1 = hh * hw_ * m, w2 = hh * lw * m, w3 = lh * hw_ * m, w4 = lh * lw * m;
          if (cx.x >= 0) {
            unsafeAtomicAdd(a.gy + cx.x + c0, w1 * g0);
            unsafeAtomicAdd(a.gy + cx.x + c0 + 32, w1 * g1);
          }
          if (cx.y >= 0) {
            unsafeAtomicAdd(a.gy + cx.y + c0, w2 * g0);
            unsafeAtomicAdd(a.gy + cx.y + c0 + 32, w2 * g1);
          }
          if (cx.z >= 0) {
            unsafeAtomicAdd(a.gy + cx.z + c0, w3 * g0);
            unsafeAtomicAdd(a.gy + cx.z + c0 + 32, w3 * g1);
          }
          if (cx.w >= 0) {
            unsafeAtomicAdd(a.gy + cx.w + c0, w4 * g0);
            unsafeAtomicAdd(a.gy + cx.w + c0 + 32, w4 * g1);
          }
        }
      }
      wave_sync_lds();
    }
  }
}

// Sampling geometry of one chunk, once per (image, tap, output pixel), for cts_bwd_bin_kernel:
// floor of the position packed as (h_low + 1) << 16 | (w_low + 1) (~0u: the sample reads
// nothing, deform_conv_cuda_kernel.cu:226) and its fractions (lh, lw).
__global__ __launch_bounds__(256) void cts_bwd_geo_kernel(BwdCtsArgs a) {
  const int HWo = a.Ho * a.Wo;
  const long long total = (long long)a.nimg * a.T * HWo;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const int hw = (int)(k % HWo);
    const long long bt = k / HWo;
    const int t = (int)(bt % a.T), bl = (int)(bt / a.T);
    const int b = a.img0 + bl;
    const int ti = t / a.kW, tj = t - ti * a.kW;
    const int ho = hw / a.Wo, wo = hw - ho * a.Wo;
    const float *op = a.offset + ((size_t)b * 2 * a.T + 2 * t) * HWo + hw;
    const float h_im = (float)(ho - a.pH + ti * a.dH) + ldg_f32(op);
    const float w_im = (float)(wo - a.pW + tj * a.dW) + ldg_f32(op + HWo);
    unsigned pk = ~0u;
    f32x2 fr = {0.f, 0.f};
    if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
      const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
      pk = ((unsigned)(h_low + 1) << 16) | (unsigned)(w_low + 1);
      fr[0] = h_im - (float)h_low;
      fr[1] = w_im - (float)w_low;
    }
    a.geo_q[k] = pk;
    a.geo_f[k] = fr;
  }
}

// Coordinate and mask gradients of one chunk when grad_Y is binned (the scatter is gone from
// this pass, so the lanes are laid out for the gathers): 16 lanes per output pixel, 4 channels
// = one 16-byte load per lane and corner, 4 pixels per wavefront, two taps' eight corner
// loads in flight; geometry from cts_bwd_geo_kernel.  Same terms as get_coordinate_weight
// (deform_conv_cuda_kernel.cu:146-188) through sum_o grad_out[o] * Y_t[corner][o].
__global__ __launch_bounds__(256) void cts_bwd_coord_kernel(BwdCtsArgs a) {
  const int tid = threadIdx.x, l = tid & 15;
  const int HWo = a.Ho * a.Wo;
  const int N = a.T * a.Cout;
  const int cb = blockIdx.y;
  const long long P = (long long)a.nimg * HWo;
  const long long stride = (long long)gridDim.x * 16;
  for (long long p = (long long)blockIdx.x * 16 + (tid >> 4); p < P; p += stride) {
    const int bl = (int)(p / HWo);
    const int hw = (int)(p - (long long)bl * HWo);
    const int b = a.img0 + bl;
    const u32x4 graw = ldg128(a.go + ((size_t)b * HWo + hw) * a.Cout + cb * 64 + 4 * l);
    const f32x4 g = __builtin_bit_cast(f32x4, graw);
    const float *ybase = a.y + (size_t)bl * a.H * a.W * N + cb * 64 + 4 * l;
    auto tap = [&](int t, f32x4 (&v)[4], float &lh, float &lw, bool &in) {
      const size_t k = ((size_t)bl * a.T + t) * HWo + hw;
      const unsigned pk = a.geo_q[k];
      const f32x2 fr = a.geo_f[k];
      lh = fr[0];
      lw = fr[1];
      in = pk != ~0u;
      const int h_low = (int)(pk >> 16) - 1, w_low = (int)(pk & 0xffffu) - 1;
      const bool hl = in && h_low >= 0, hh = in && h_low + 1 <= a.H - 1;
      const bool wl = w_low >= 0, wh = w_low + 1 <= a.W - 1;
      const float *c1 = ybase + ((long long)h_low * a.W + w_low) * N + t * a.Cout;
      const u32x4 z = {0u, 0u, 0u, 0u};
      v[0] = __builtin_bit_cast(f32x4, hl && wl ? ldg128(c1) : z);
      v[1] = __builtin_bit_cast(f32x4, hl && wh ? ldg128(c1 + N) : z);
      v[2] = __builtin_bit_cast(f32x4, hh && wl ? ldg128(c1 + (size_t)a.W * N) : z);
      v[3] = __builtin_bit_cast(f32x4, hh && wh ? ldg128(c1 + (size_t)a.W * N + N) : z);
    };
    auto finish = [&](int t, const f32x4 (&v)[4], float lh, float lw, bool in) {
      const float hh = 1.f - lh, hw_ = 1.f - lw;
      float sh = 0.f, sw = 0.f, sm = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sh += g[c] * (-hw_ * v[0][c] - lw * v[1][c] + hw_ * v[2][c] + lw * v[3][c]);
        sw += g[c] * (-hh * v[0][c] + hh * v[1][c] - lh * v[2][c] + lh * v[3][c]);
        sm += g[c] * (hh * hw_ * v[0][c] + hh * lw * v[1][c] + lh * hw_ * v[2][c] + lh * lw * v[3][c]);
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        sh += __shfl_xor(sh, o);
        sw += __shfl_xor(sw, o);
        sm += __shfl_xor(sm, o);
      }
      if (l == 0 && in) {
        const float m = a.mask ? ldg_f32(a.mask + ((size_t)b * a.T + t) * HWo + hw) : 1.f;
        float *q = a.goff + ((size_t)b * 2 * a.T + 2 * t) * HWo + hw;
        unsafeAtomicAdd(q, m * sh);
        unsafeAtomicAdd(q + HWo, m * sw);
        if (a.gmask) unsafeAtomicAdd(a.gmask + ((size_t)b * a.T + t) * HWo + hw, sm);
      }
    };
    int t = 0;
    for (; t + 1 < a.T; t += 2) {
      f32x4 va[4], vb[4];
      float lha, lwa, lhb, lwb;
      bool ina, inb;
      tap(t, va, lha, lwa, ina);
      tap(t + 1, vb, lhb, lwb, inb);
      finish(t, va, lha, lwa, ina);
      finish(t + 1, vb, lhb, lwb, inb);
    }
    if (t < a.T) {
      f32x4 va[4];
      float lha, lwa;
      bool ina;
      tap(t, va, lha, lwa, ina);
      finish(t, va, lha, lwa, ina);
    }
  }
}

// grad_Y without global atomics.  The L2 retires about one float add per channel and clock
// (tools/atomicbw.hip: 300 G adds/s whatever the locality), which is what the scatter above
// runs at; LDS adds are two orders of magnitude cheaper.  One workgroup owns BIN_Q consecutive
// input pixels q of one image, one tap and one 64-channel block as a float tile in LDS.  It
// walks ALL output pixels of the image: every thread forms one pixel's sampling position and
// appends the corners that land in the tile to an LDS list (pixel, tile row, weight * mask);
// half wavefronts then take the entries in turn, read the pixel's grad_out run (lane l owns
// channels l and l + 32) and add into the tile.  The tile leaves once, as plain 16-byte
// stores that cover every (q, tap) of the chunk: grad_Y needs no memset.  Each part scans the
// whole image, so the form is used while an image has at most BIN_MAX_PARTS parts.  A list that
// fills up (every sample of the image landing in one tile) makes the thread add its run itself
// with LDS atomics.
constexpr int BIN_Q = 128;                // power of two <= 128: a list entry packs pixel << 7 | row
constexpr int BIN_THREADS = 512;
constexpr int BIN_CAP = 2560;             // list entries of 8 bytes (tile + lists = 52.1 KB: three workgroups per CU)
constexpr int BIN_MAX_PARTS = 512;         // 65536 pixels: still ahead of the atomics there (160x400: 2.38 vs 2.65 ms)

__global__ __launch_bounds__(BIN_THREADS) void cts_bwd_bin_kernel(BwdCtsArgs a, int parts) {
  __shared__ __attribute__((aligned(16))) float s_tile[BIN_Q * 64];
  __shared__ int s_ep[BIN_CAP];           // output pixel << 7 | tile row
  __shared__ float s_ew[BIN_CAP];
  __shared__ int s_cnt[BIN_THREADS / 32];
  const int tid = threadIdx.x, l = tid & 31;
  const int hwv = __builtin_amdgcn_readfirstlane(tid >> 6) * 2 + ((tid >> 5) & 1);   // half wavefront
  const int HW = a.H * a.W, HWo = a.Ho * a.Wo;
  const int N = a.T * a.Cout;
  int task = blockIdx.x;
  const int part = task % parts;
  task /= parts;
  const int t = task % a.T;
  const int bl = task / a.T;
  const int b = a.img0 + bl;
  const int cb = blockIdx.y;
  const int q0 = part * BIN_Q;

  for (int k = tid; k < BIN_Q * 64 / 4; k += BIN_THREADS)
    reinterpret_cast<f32x4 *>(s_tile)[k] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (tid < BIN_THREADS / 32) s_cnt[tid] = 0;
  __syncthreads();

  const float *gob = a.go + (size_t)b * HWo * a.Cout + cb * 64;
  const unsigned *gq = a.geo_q + ((size_t)bl * a.T + t) * HWo;
  const f32x2 *gf = a.geo_f + ((size_t)bl * a.T + t) * HWo;
  const float *mb = a.mask ? a.mask + ((size_t)b * a.T + t) * HWo : nullptr;
  constexpr int NH = BIN_THREADS / 32, LCAP = BIN_CAP / NH, DU = 8;
  // one pass over the image's output pixels: the packed corner of cts_bwd_geo_kernel decides
  // in a few instructions whether a sample touches this tile (every tile scans every pixel)
  for (int base = 0; base < HWo; base += 4 * BIN_THREADS) {
    unsigned pk[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int hw = base + u * BIN_THREADS + tid;
      pk[u] = hw < HWo ? gq[hw] : ~0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int hw = base + u * BIN_THREADS + tid;
      if (pk[u] == ~0u) continue;
      const int h_low = (int)(pk[u] >> 16) - 1, w_low = (int)(pk[u] & 0xffffu) - 1;
      const int qa = h_low * a.W + w_low - q0;          // tile row of corner 1
      if (qa + a.W + 1 < 0 || qa >= BIN_Q) continue;
      const f32x2 fr = gf[hw];
      const float lh = fr[0], lw = fr[1];
      const float hh = 1.f - lh, hw_ = 1.f - lw;
      const float m = mb ? ldg_f32(mb + hw) : 1.f;
      const bool hl = h_low >= 0, hhi = h_low + 1 <= a.H - 1;
      const bool wl = w_low >= 0, whi = w_low + 1 <= a.W - 1;
      auto put = [&](bool ok, int r, float w) {
        if (ok && r >= 0 && r < BIN_Q) {
          const int own = r & (NH - 1);          // the half wavefront that owns tile row r
          const int slot = atomicAdd(&s_cnt[own], 1);
          if (slot < LCAP) {
            s_ep[own * LCAP + slot] = hw << 7 | r;
            s_ew[own * LCAP + slot] = w * m;
          } else {
            // list full (far more samples than usual land in this tile): add the run here
            const float *gp = gob + (size_t)hw * a.Cout;
            for (int c = 0; c < 64; ++c) unsafeAtomicAdd(&s_tile[r * 64 + c], w * m * ldg_f32(gp + c));
          }
        }
      };
      put(hl && wl, qa, hh * hw_);
      put(hl && whi, qa + 1, hh * lw);
      put(hhi && wl, qa + a.W, lh * hw_);
      put(hhi && whi, qa + a.W + 1, lh * lw);
    }
  }
  __syncthreads();
  // Half wavefront hwv owns the tile rows r with r % NH == hwv and walks its own list: plain
  // read-modify-writes, in program order, on rows nobody else touches (the inline adds above
  // ended at the barrier).  Eight entries' grad_out runs are in flight at a time.
  const int cnt = s_cnt[hwv] < LCAP ? s_cnt[hwv] : LCAP;
  const int lb = hwv * LCAP;
  int e = 0;
  for (; e + DU <= cnt; e += DU) {
    float g0[DU], g1[DU], w[DU];
    int r[DU];
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      const int pr = s_ep[lb + e + u];
      const float *gp = gob + (size_t)(pr >> 7) * a.Cout + l;
      g0[u] = ldg_f32(gp);
      g1[u] = ldg_f32(gp + 32);
      r[u] = pr & (BIN_Q - 1);
      w[u] = s_ew[lb + e + u];
    }
#pragma unroll
    for (int u = 0; u < DU; ++u) {
      s_tile[r[u] * 64 + l] += w[u] * g0[u];
      s_tile[r[u] * 64 + l + 32] += w[u] * g1[u];
    }
  }
  for (; e < cnt; ++e) {
    const int pr = s_ep[lb + e];
    const float *gp = gob + (size_t)(pr >> 7) * a.Cout + l;
    const float w = s_ew[lb + e];
    const int r = pr & (BIN_Q - 1);
    s_tile[r * 64 + l] += w * ldg_f32(gp);
    s_tile[r * 64 + l + 32] += w * ldg_f32(gp + 32);
  }
  __syncthreads();

  // rows of the tile that exist in the image leave as whole 256-byte runs
  const int rows = HW - q0 < BIN_Q ? HW - q0 : BIN_Q;
  float *dst = a.gy + ((size_t)bl * HW + q0) * N + t * a.Cout + cb * 64;
  for (int k = tid; k < rows * 16; k += BIN_THREADS) {
    const int r = k >> 4, c4 = k & 15;
    *reinterpret_cast<f32x4 *>(dst + (size_t)r * N + c4 * 4) =
        *reinterpret_cast<const f32x4 *>(s_tile + r * 64 + c4 * 4);
  }
}

// gw[n][c] += sum_q gy[q][n] * x[q][c]   (gy [M][N], x [M][Cp], gw [N][Cp]; N, Cp % 64 == 0)
// blockIdx.x = (n block, c block) of 64 x 64, blockIdx.y = share of the pixels; the four
// wavefronts of a workgroup split its pixels and meet in LDS before one atomic flush.
__global__ __launch_bounds__(256) void cts_bwd_wgrad_kernel(const float *__restrict__ gy,
                                                            const float *__restrict__ x,
                                                            float *__restrict__ gw, long long M,
                                                            int N, int Cp) {
  __shared__ float s_d[64][65];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int ncb = Cp / 64;
  const int bn = blockIdx.x / ncb, bc = blockIdx.x - bn * ncb;
  for (int k = tid; k < 64 * 65; k += 256) (&s_d[0][0])[k] = 0.f;
  __syncthreads();
  // pixel pairs of this workgroup, dealt to its wavefronts in turn
  const long long npairs = (M + 1) / 2;
  const long long per = (npairs + gridDim.y - 1) / gridDim.y;
  const long long m_beg = (long long)blockIdx.y * per;
  const long long m_end = m_beg + per < npairs ? m_beg + per : npairs;
  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
  const float *gcol = gy + bn * 64 + i;
  const float *xcol = x + bc * 64 + i;
#pragma unroll 4
  for (long long m = m_beg + wid; m < m_end; m += 4) {
    const long long q = 2 * m + h;
    float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
    if (q < M) {
      a0 = ldg_f32(gcol + q * N);
      a1 = ldg_f32(gcol + q * N + 32);
      b0 = ldg_f32(xcol + q * Cp);
      b1 = ldg_f32(xcol + q * Cp + 32);
    }
    // D[n][c] += gy[q][n] * x[q][c], k = the pixel pair
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = u * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        unsafeAtomicAdd(&s_d[row][v * 32 + i], acc[u][v][r]);
      }
  __syncthreads();
  for (int k = tid; k < 64 * 64; k += 256) {
    const int row = k >> 6, col = k & 63;
    unsafeAtomicAdd(gw + (size_t)(bn * 64 + row) * Cp + bc * 64 + col, s_d[row][col]);
  }
}

// The same product on the bf16 matrix cores in three passes over high/low splits (the
// float32 MFMA runs at the vector rate, and 64 x 64 blocks re-read gy and x 4 + 7 times).
// One workgroup of 8 wavefronts owns NH*NTW*32 rows (n) x (8/NH)*32 columns (c) of gw and a
// share of the pixels; wavefront w keeps NTW 32 x 32 accumulators (c tile w % (8/NH), n part
// w / (8/NH)).  NH = 2: 2*NTW*32 x 128 (gy is read Cp/128 times, x N/(2*NTW*32) times);
// NH = 1: NTW*32 x 256 (gy Cp/256 times, x N/(NTW*32) times) - at the LFA's 448 x 256 that is
// gy once + x twice (491 MB for 32 images) instead of gy twice + x once (589 MB).
// K = pixels, 16 per step: 288 threads each fetch 8 pixels x 4 columns of gy or x as float32
// (whole 16-byte loads, a pixel row contiguous across lanes), split them, and park them in
// LDS pixel-major per column ([plane][k half][column][8 pixels] bf16: the 16 bytes a lane
// reads are one MFMA operand, 32 lanes read 512 contiguous bytes); two buffers, one barrier
// per step.  At the end the accumulators go to the share's own plane of partial sums (128-byte
// runs; `part`), or, without one, into gw through atomics.
constexpr int WG_WAVES = 8;
constexpr int WG_KS = 16;

template <int NTW, int NH>
__global__ __launch_bounds__(WG_WAVES * 64) void cts_bwd_wgrad_split_kernel(
    const float *__restrict__ gy, const float *__restrict__ x, float *__restrict__ gw,
    long long M, int N, int Cp, float *__restrict__ part, int first) {
  constexpr int BN = NH * NTW * 32;
  constexpr int CT = WG_WAVES / NH;             // c tiles of 32 per workgroup
  constexpr int WG_BC = CT * 32;
  constexpr int NCOLS = BN + WG_BC;
  constexpr int NQ = NCOLS / 4;                 // column quads
  constexpr int HALF = NCOLS * 16;              // bytes of one k half of one plane
  constexpr int BUF = 4 * HALF;                 // hi (two halves), lo (two halves)
  static_assert(2 * NQ <= WG_WAVES * 64, "one fetch task per thread");
  __shared__ __attribute__((aligned(16))) unsigned char s_t[2 * BUF];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const int ncb = Cp / WG_BC;
  const int bn = blockIdx.x / ncb, bc = blockIdx.x - bn * ncb;
  const int nb0 = bn * BN, cb0 = bc * WG_BC;
  const int ct = wid % CT, nh = wid / CT;

  const long long nks = (M + WG_KS - 1) / WG_KS;
  const long long per = (nks + gridDim.y - 1) / gridDim.y;
  const long long ks_beg = (long long)blockIdx.y * per;
  const long long ks_end = ks_beg + per < nks ? ks_beg + per : nks;
  // no pixels for this share: the whole workgroup leaves together - unless its plane of partial
  // sums has never been written, then it writes the zeros it holds
  const bool idle = ks_beg >= ks_end;
  if (idle && !(part && first)) return;

  // fetch task: pixel octet po of the step, column quad cq (gy columns first, then x)
  const bool fetcher = tid < 2 * NQ;
  const int po = tid >= NQ ? 1 : 0, cq = tid - po * NQ;
  const bool from_gy = 4 * cq < BN;
  const float *src = from_gy ? gy + nb0 + 4 * cq : x + cb0 + (4 * cq - BN);
  const long long ld = from_gy ? N : Cp;
  // two register sets: raw holds the step that is parked next, ahead the one after it, so a
  // step's rows have a whole step to arrive (one loop body, the sets rotate by moves: two
  // unrolled bodies spilled)
  u32x4 raw[8], ahead[8];
  auto fetch = [&](u32x4 (&raw)[8], long long ks) {
    const long long q0 = ks * WG_KS + po * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const long long q = q0 + j;
      const u32x4 v = ldg128(src + (q < M ? q : M - 1) * ld);
      const u32x4 z = {0u, 0u, 0u, 0u};
      raw[j] = q < M ? v : z;
    }
  };
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  auto park = [&](int buf) {
    unsigned char *dst = s_t + buf * BUF + po * HALF + (4 * cq) * 16;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned int hi[4], lo[4];
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        const float a0 = __uint_as_float(raw[2 * jj][e]);
        const float a1 = __uint_as_float(raw[2 * jj + 1][e]);
        const bf16x2 ph = {(__bf16)a0, (__bf16)a1};
        const unsigned int uh = __builtin_bit_cast(unsigned int, ph);
        const float r0 = a0 - __uint_as_float(uh << 16);
        const float r1 = a1 - __uint_as_float(uh & 0xffff0000u);
        const bf16x2 pl = {(__bf16)r0, (__bf16)r1};
        hi[jj] = uh;
        lo[jj] = __builtin_bit_cast(unsigned int, pl);
      }
      const u32x4 h4 = {hi[0], hi[1], hi[2], hi[3]};
      const u32x4 l4 = {lo[0], lo[1], lo[2], lo[3]};
      *reinterpret_cast<u32x4 *>(dst + e * 16) = h4;
      *reinterpret_cast<u32x4 *>(dst + 2 * HALF + e * 16) = l4;
    }
  };

  f32x16 acc[NTW];
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  if (fetcher && !idle) {
    fetch(raw, ks_beg);
    park(0);
    if (ks_beg + 1 < ks_end) fetch(raw, ks_beg + 1);
    if (ks_beg + 2 < ks_end) fetch(ahead, ks_beg + 2);
  }
  __syncthreads();
#pragma unroll 1
  for (long long ks = ks_beg; ks < ks_end; ++ks) {
    const int buf = (int)(ks - ks_beg) & 1;
    const unsigned char *b = s_t + buf * BUF + h * HALF;
    const bf16x8 xh = *reinterpret_cast<const bf16x8 *>(b + (BN + ct * 32 + i) * 16);
    const bf16x8 xl = *reinterpret_cast<const bf16x8 *>(b + 2 * HALF + (BN + ct * 32 + i) * 16);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
      const int col = (nh * NTW + t) * 32 + i;
      const bf16x8 gh = *reinterpret_cast<const bf16x8 *>(b + col * 16);
      const bf16x8 gl = *reinterpret_cast<const bf16x8 *>(b + 2 * HALF + col * 16);
      // rows = n (grad_Y columns), columns = c (x channels); small terms first
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gl, xh, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xl, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(gh, xh, acc[t], 0, 0, 0);
    }
    if (fetcher && ks + 1 < ks_end) park(buf ^ 1);
    if (fetcher && ks + 2 < ks_end) {
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = ahead[j];
    }
    if (fetcher && ks + 3 < ks_end) fetch(ahead, ks + 3);
    __syncthreads();
  }

  // lane (i, h), register r of tile t: row (r & 3) + 8 (r >> 2) + 4 h, column i
  const size_t at = (size_t)(nb0 + nh * NTW * 32 + 4 * h) * Cp + cb0 + ct * 32 + i;
  if (part) {
    // this share's own plane: plain stores (the first chunk of a call) or load-add-store
    // (later chunks; the launches of a call are ordered on its stream and a plane has one
    // owner per launch), summed in plane order by bwd_reduce_unpack_weight_kernel
    float *out = part + (size_t)blockIdx.y * N * Cp + at;
#pragma unroll
    for (int t = 0; t < NTW; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float *dst = out + (size_t)(t * 32 + (r & 3) + 8 * (r >> 2)) * Cp;
        *dst = first ? acc[t][r] : *dst + acc[t][r];
      }
    return;
  }
  float *out = gw + at;
#pragma unroll
  for (int t = 0; t < NTW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      unsafeAtomicAdd(out + (size_t)(t * 32 + (r & 3) + 8 * (r >> 2)) * Cp, acc[t][r]);
}

// Launches the split kernel where its tiling fits (N a multiple of 64 * NTW for an
// instantiated NTW, Cp a multiple of 128); false = use cts_bwd_wgrad_kernel.
struct WgradGeometry {
  bool tall;
  int ntw, gx;
  long long split;   // shares of the pixels = planes of partial sums
};

// false = the split kernel's tiling does not fit (use cts_bwd_wgrad_kernel)
bool wgrad_split_geometry(long long M, int N, int Cp, WgradGeometry *g) {
  if (getenv("GANET_BWD_WGRAD_F32")) return false;   // A/B knob
  if (N % 64 || Cp % 128 || M < 1) return false;
  // tall tiles (NH = 1: NTW*32 rows x 256 columns) where gy is the wider matrix and the shape
  // allows: gy is then read once per 256 columns of x instead of once per 128
  g->tall = Cp % 256 == 0 && N >= Cp && !getenv("GANET_BWD_WGRAD_WIDE");
  const int m = g->tall ? N / 32 : N / 64;        // row blocks of 32 (tall) or 64 (two halves)
  g->ntw = 0;
  for (int cand : {7, 9, 5, 4, 3, 2, 1})
    if (m % cand == 0) {
      g->ntw = cand;
      break;
    }
  if (!g->ntw) return false;
  g->gx = (m / g->ntw) * (Cp / (g->tall ? 256 : 128));
  const long long nks = (M + WG_KS - 1) / WG_KS;
  g->split = 256 / g->gx;                           // one workgroup per CU
  if (g->split > nks / 8) g->split = nks / 8;       // >= 8 steps between flushes
  if (g->split < 1) g->split = 1;
  return true;
}

// part: planes of partial sums ([split][N][Cp], `first` = not written yet in this call), or
// nullptr = add into gw through atomics.  *planes = the shares of this launch.
bool launch_wgrad_split(const float *gy, const float *x, float *gw, long long M, int N, int Cp,
                        float *part, int first, long long *planes, hipStream_t s) {
  WgradGeometry g;
  if (!wgrad_split_geometry(M, N, Cp, &g)) return false;
  const bool tall = g.tall;
  const int ntw = g.ntw, gx = g.gx;
  const long long split = g.split;
  if (planes) *planes = split;
  const dim3 grid((unsigned)gx, (unsigned)split), block(WG_WAVES * 64);
#define GANET_WGRAD_CASE(n)                                                                       \
  case n:                                                                                         \
    if (tall)                                                                                     \
      hipLaunchKernelGGL((cts_bwd_wgrad_split_kernel<n, 1>), grid, block, 0, s, gy, x, gw, M, N, Cp, part, first); \
    else                                                                                          \
      hipLaunchKernelGGL((cts_bwd_wgrad_split_kernel<n, 2>), grid, block, 0, s, gy, x, gw, M, N, Cp, part, first); \
    break;
  switch (ntw) {
    GANET_WGRAD_CASE(1)
    GANET_WGRAD_CASE(2)
    GANET_WGRAD_CASE(3)
    GANET_WGRAD_CASE(4)
    GANET_WGRAD_CASE(5)
    GANET_WGRAD_CASE(7)
    default:
      if (tall)
        hipLaunchKernelGGL((cts_bwd_wgrad_split_kernel<9, 1>), grid, block, 0, s, gy, x, gw, M, N, Cp, part, first);
      else
        hipLaunchKernelGGL((cts_bwd_wgrad_split_kernel<9, 2>), grid, block, 0, s, gy, x, gw, M, N, Cp, part, first);
      break;
  }
#undef GANET_WGRAD_CASE
  return true;
}

// The commuted form's weight views in one launch (they were four: pack, transpose, two
// splits, ~5 us each of launch latency): wp[n][c] float32, its transpose wqt[c][n], and the
// high / low bf16 planes of both (wsp: hi[n][c], lo[n][c], hi[c][n], lo[c][n]).
template <typename TS>
__global__ __launch_bounds__(256) void bwd_weight_views_kernel(
    const TS *__restrict__ w, float *__restrict__ wp, float *__restrict__ wqt,
    unsigned short *__restrict__ wsp, int Cout, int Cin, int T, int Cp, int groups) {
  const int cig = Cin / groups, cog = Cout / groups;
  const int N = T * Cout;
  const long long total = (long long)N * Cp;
  for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total;
       k += (long long)gridDim.x * 256) {
    const int cc = (int)(k % Cp);
    const int n = (int)(k / Cp);
    const int o = n % Cout, t = n / Cout;
    const int cg = cc - (o / cog) * cig;
    const float v = (cc < Cin && cg >= 0 && cg < cig) ? to_f32(w[((size_t)o * cig + cg) * T + t]) : 0.f;
    const __bf16 hi = (__bf16)v;
    const float hf = __uint_as_float(((unsigned)__builtin_bit_cast(unsigned short, hi)) << 16);
    const __bf16 lo = (__bf16)(v - hf);
    const unsigned short uh = __builtin_bit_cast(unsigned short, hi);
    const unsigned short ul = __builtin_bit_cast(unsigned short, lo);
    const long long kt = (long long)cc * N + n;
    wp[k] = v;
    wqt[kt] = v;
    wsp[k] = uh;
    wsp[total + k] = ul;
    wsp[2 * total + kt] = uh;
    wsp[3 * total + kt] = ul;
  }
}

struct BwdPlan {
  int Ho, Wo, T, Cp;
  bool copy_x, copy_go;
  size_t xh_bytes, gxh_bytes, go_bytes, wp_bytes, gwp_bytes;
  // commuted form
  bool commuted;
  int nb;                 // images per chunk
  size_t y_bytes, gy_bytes, gxc_bytes, wqt_bytes, wsp_bytes, geo_bytes;
  size_t wpart_bytes;   // the split weight-gradient kernel's planes of partial sums
  size_t gb_bytes;   // float32 bias gradient of a 16-bit call
  bool overwrite_x;  // GANET_DCN_BWD_OVERWRITE_X: grad_x is written, not added into
  size_t total() const {
    return xh_bytes + gxh_bytes + go_bytes + wp_bytes + gwp_bytes + y_bytes + gy_bytes +
           gxc_bytes + wqt_bytes + wsp_bytes + geo_bytes + wpart_bytes + gb_bytes;
  }
};

int make_bwd_plan(const ganet_dcn_params *p_in, BwdPlan *pl) {
  if (!p_in) return GANET_ERR_NULL;
  ganet_dcn_params q = *p_in;   // the forward's GANET_DCN_ALGO_KEEP_Y bit means nothing here
  if (q.algo < 0 || (q.algo & ~(0xff | GANET_DCN_ALGO_KEEP_Y | GANET_DCN_BWD_OVERWRITE_X)))
    return GANET_ERR_BAD_SHAPE;
  pl->overwrite_x = (q.algo & GANET_DCN_BWD_OVERWRITE_X) != 0;
  q.algo &= 0xff;
  if (q.algo > 2) return GANET_ERR_BAD_SHAPE;
  const ganet_dcn_params *p = &q;
  if (p->B <= 0 || p->Cin <= 0 || p->H <= 0 || p->W <= 0 || p->Cout <= 0)
    return GANET_ERR_BAD_SHAPE;
  if (p->kH <= 0 || p->kW <= 0 || p->sH <= 0 || p->sW <= 0 || p->dH <= 0 || p->dW <= 0 ||
      p->pH < 0 || p->pW < 0)
    return GANET_ERR_BAD_SHAPE;
  if (p->groups <= 0 || p->deformable_groups <= 0) return GANET_ERR_BAD_SHAPE;
  if (p->Cin % p->groups || p->Cout % p->groups || p->Cin % p->deformable_groups)
    return GANET_ERR_BAD_SHAPE;
  if (p->dtype != GANET_F32 && p->dtype != GANET_BF16 && p->dtype != GANET_F16)
    return GANET_ERR_BAD_SHAPE;
  // 16-bit tensors (the reference dispatches half here too, deform_conv_cuda_kernel.cu:259 and
  // the col2im launchers): x, grad_out and the weights are widened to float32 copies in the
  // workspace, the gradients are formed in float32 and rounded once on the way out
  const bool io16 = p->dtype != GANET_F32;
  if (p->offset_is_points || p->fuse_relu || p->weight_is_packed) return GANET_ERR_UNSUPPORTED;
  if (!ganet_dcn_ranges_ok(p)) return GANET_ERR_UNSUPPORTED;
  pl->Ho = ganet_dcn_out_size(p->H, p->pH, p->dH, p->kH, p->sH);
  pl->Wo = ganet_dcn_out_size(p->W, p->pW, p->dW, p->kW, p->sW);
  if (pl->Ho < 1 || pl->Wo < 1) return GANET_ERR_BAD_SHAPE;
  pl->T = p->kH * p->kW;
  const int dg = p->deformable_groups;
  // channel tiles of 32 must not straddle deformable groups
  if (dg > 1 && (p->Cin / dg) % 32) return GANET_ERR_UNSUPPORTED;
  pl->Cp = (p->Cin + 31) / 32 * 32;
  const size_t npix_in = (size_t)p->B * p->H * p->W;
  if ((long long)p->B * pl->Ho >= (1ll << 31)) return GANET_ERR_UNSUPPORTED;
  const size_t npix_out = (size_t)p->B * pl->Ho * pl->Wo;
  if (npix_in * pl->Cp >= 0x7fffffffull || npix_out >= 0x7fffffffull)
    return GANET_ERR_UNSUPPORTED;  // 32-bit row indices in the sample table
  pl->copy_x = p->x_layout != GANET_NHWC || pl->Cp != p->Cin || io16;
  pl->copy_go = p->out_layout != GANET_NHWC || io16;
  pl->gb_bytes = io16 ? ganet_align_up((size_t)p->Cout * 4, 256) : 0;
  pl->xh_bytes = pl->copy_x ? ganet_align_up(npix_in * pl->Cp * 4, 256) : 0;
  pl->gxh_bytes = pl->xh_bytes;
  pl->go_bytes = pl->copy_go ? ganet_align_up(npix_out * p->Cout * 4, 256) : 0;
  pl->wp_bytes = ganet_align_up((size_t)pl->T * p->Cout * pl->Cp * 4, 256);
  pl->gwp_bytes = pl->wp_bytes;
  // The commuted form applies under the forward's contract-then-sample conditions (one
  // deformable group, stride 1, Cout-wide rows at most half as wide as Cin-wide ones) and
  // needs whole 64-blocks on both sides of its GEMMs; p->algo: 0 automatic, 1 direct,
  // 2 commuted (GANET_ERR_UNSUPPORTED where it does not apply).
  const bool can = dg == 1 && p->sH == 1 && p->sW == 1 && p->Cout % 64 == 0 &&
                   pl->Cp % 64 == 0 && pl->T <= 64 &&
                   (size_t)p->H * p->W * pl->T * p->Cout < 0x7fffffffull / 4;
  if (p->algo == 2 && !can) return GANET_ERR_UNSUPPORTED;
  pl->commuted = can && (p->algo == 2 || (p->algo == 0 && 2 * p->Cout <= pl->Cp));
  pl->nb = 0;
  pl->y_bytes = pl->gy_bytes = pl->gxc_bytes = pl->wqt_bytes = pl->wsp_bytes = pl->geo_bytes = 0;
  pl->wpart_bytes = 0;
  if (pl->commuted) {
    const size_t N = (size_t)pl->T * p->Cout;
    const size_t per_img = (size_t)p->H * p->W * N * 4;
    // Y and grad_Y of a chunk: with the scatter binned in LDS, fewer and larger launches beat
    // keeping both inside the Infinity Cache (B=32: 96 MB 1.05 ms, 128 MB 0.95, 256 MB 0.91)
    size_t budget = 256ull << 20;
    if (const char *e = getenv("GANET_BWD_BUDGET_MB")) {   // A/B knob, read per call: the call sizes
      const long long mb = atoll(e);                       // its own need and refuses a smaller workspace
      if (mb > 0) budget = (size_t)mb << 20;
    }
    size_t nb = budget / per_img;
    const size_t lim = (0x7fffffffull / 4) / ((size_t)p->H * p->W * N);   // 31-bit float indices
    if (nb > lim) nb = lim;
    if (nb < 1) nb = 1;
    if (nb > (size_t)p->B) nb = p->B;
    const size_t chunks = ((size_t)p->B + nb - 1) / nb;
    nb = ((size_t)p->B + chunks - 1) / chunks;
    pl->nb = (int)nb;
    pl->y_bytes = ganet_align_up(nb * per_img, 256);
    pl->gy_bytes = pl->y_bytes;
    pl->gxc_bytes = ganet_align_up(nb * (size_t)p->H * p->W * pl->Cp * 4, 256);
    pl->wqt_bytes = ganet_align_up(N * pl->Cp * 4, 256);
    pl->wsp_bytes = 2 * pl->wqt_bytes;   // Wq and its transpose, two bf16 planes each
    // sampling geometry of a chunk for the binned grad_Y: packed corner (4 B) + fractions (8 B)
    pl->geo_bytes = ganet_align_up(nb * (size_t)pl->T * pl->Ho * pl->Wo * 4, 256) +
                    ganet_align_up(nb * (size_t)pl->T * pl->Ho * pl->Wo * 8, 256);
    // grad_x is built per chunk in gxc and added to the caller's tensor: no full-size copy
    pl->gxh_bytes = 0;
    // the weight-gradient GEMM's partial sums: one [N][Cp] plane per share of the pixels (the
    // first chunk is the largest, so it has the most shares); GANET_BWD_WGRAD_ATOMIC=1 keeps
    // the atomic flush into one plane (A/B knob, read per call like the other backward knobs)
    WgradGeometry g;
    if (!getenv("GANET_BWD_WGRAD_ATOMIC") &&
        wgrad_split_geometry((long long)nb * p->H * p->W, (int)N, pl->Cp, &g))
      pl->wpart_bytes = ganet_align_up((size_t)g.split * N * pl->Cp * 4, 256);
  }
  return GANET_OK;
}

}  // namespace

}  // namespace ganet_dcn

using namespace ganet_dcn;

// runs `...` with TIO = the element type of the caller's tensors
#define GANET_IO_DISPATCH(dt, ...)      \
  do {                                  \
    if ((dt) == GANET_F32) {            \
      typedef float TIO;                \
      __VA_ARGS__;                      \
    } else if ((dt) == GANET_BF16) {    \
      typedef bf16_t TIO;               \
      __VA_ARGS__;                      \
    } else {                            \
      typedef f16_t TIO;                \
      __VA_ARGS__;                      \
    }                                   \
  } while (0)

extern "C" size_t ganet_deform_conv_backward_workspace_bytes(const ganet_dcn_params *p) {
  BwdPlan pl;
  if (make_bwd_plan(p, &pl) != GANET_OK) return 0;
  return pl.total();
}

namespace {
int backward_impl(const void *x, const float *offset, const float *mask, const void *weight,
                  const void *grad_out, void *grad_x, float *grad_offset, float *grad_mask,
                  void *grad_weight, void *grad_bias, const ganet_dcn_params *p, void *workspace,
                  size_t workspace_bytes, ganet_stream_t stream, const float *y_saved) {
  if (!x || !offset || !weight || !grad_out || !p || !workspace) return GANET_ERR_NULL;
  const bool need_in = grad_x != nullptr || grad_offset != nullptr;
  if (need_in && (!grad_x || !grad_offset)) return GANET_ERR_NULL;  // computed together
  if (grad_mask && !mask) return GANET_ERR_NULL;
  if (mask && need_in && !grad_mask) return GANET_ERR_NULL;
  BwdPlan pl;
  const int rc = make_bwd_plan(p, &pl);
  if (rc != GANET_OK) return rc;
  const size_t need = pl.total();
  if (workspace_bytes < need) return GANET_ERR_WORKSPACE;
  // the forward's per-tap products: float32 tensors on the commuted form only (they are the
  // products this function would otherwise form again from x and the weights)
  if (y_saved && (!pl.commuted || p->dtype != GANET_F32)) return GANET_ERR_UNSUPPORTED;
  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  float *xh = reinterpret_cast<float *>(ws);
  float *gxh = reinterpret_cast<float *>(ws + pl.xh_bytes);
  float *goh = reinterpret_cast<float *>(ws + pl.xh_bytes + pl.gxh_bytes);
  float *wp = reinterpret_cast<float *>(ws + pl.xh_bytes + pl.gxh_bytes + pl.go_bytes);
  float *gwp = reinterpret_cast<float *>(ws + pl.xh_bytes + pl.gxh_bytes + pl.go_bytes + pl.wp_bytes);

  const int HW = p->H * p->W, HWo = pl.Ho * pl.Wo, T = pl.T;
  const long long npix_in = (long long)p->B * HW, P = (long long)p->B * HWo;
  const float *xf = static_cast<const float *>(x);          // float32 calls only where used as is
  const float *gof = static_cast<const float *>(grad_out);
  const int dt = p->dtype;
  auto blocks_for = [](long long n) {
    long long b = (n + 255) / 256;
    return (unsigned)(b > 65535 ? 65535 : (b < 1 ? 1 : b));
  };

  // GANET_DCN_BWD_OVERWRITE_X: where the commuted form writes the caller's channel-last
  // float32 grad_x straight from its GEMM every element is stored exactly once (no zero-fill,
  // no read of old values); every other path adds, so the tensor is cleared here first
  const bool direct_gx = pl.commuted && need_in && pl.Cp <= 256 && p->x_layout == GANET_NHWC &&
                         pl.Cp == p->Cin && p->dtype == GANET_F32;
  if (pl.overwrite_x && need_in && !direct_gx &&
      hipMemsetAsync(grad_x, 0, (size_t)npix_in * p->Cin * (p->dtype == GANET_F32 ? 4 : 2), s) !=
          hipSuccess)
    return GANET_ERR_HIP;

  // channel-last views of x and grad_out
  if (pl.copy_x) {
    if (p->x_layout == GANET_NHWC) {
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_pad_rows_kernel<TIO>,
                                               dim3(blocks_for(npix_in * pl.Cp)), dim3(256), 0, s,
                                               static_cast<const TIO *>(x), xh, npix_in, p->Cin,
                                               pl.Cp));
    } else {
      dim3 grid((HW + 63) / 64, (pl.Cp + 63) / 64, p->B);
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_nchw_to_nhwc_kernel<TIO>, grid, dim3(256), 0, s,
                                               static_cast<const TIO *>(x), xh, p->Cin, HW, pl.Cp));
    }
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }
  if (pl.copy_go) {
    if (p->out_layout == GANET_NHWC) {   // 16-bit channel-last grad_out: widened only
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_pad_rows_kernel<TIO>,
                                               dim3(blocks_for(P * p->Cout)), dim3(256), 0, s,
                                               static_cast<const TIO *>(grad_out), goh, P, p->Cout,
                                               p->Cout));
    } else {
      dim3 grid((HWo + 63) / 64, (p->Cout + 63) / 64, p->B);
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_nchw_to_nhwc_kernel<TIO>, grid, dim3(256), 0, s,
                                               static_cast<const TIO *>(grad_out), goh, p->Cout,
                                               HWo, p->Cout));
    }
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }

  BwdArgs a;
  a.xh = pl.copy_x ? xh : xf;
  a.offset = offset;
  a.mask = mask;
  a.wp = wp;
  a.go = pl.copy_go ? goh : gof;
  a.gxh = pl.copy_x ? gxh : static_cast<float *>(grad_x);
  a.goff = grad_offset;
  a.gmask = grad_mask;
  a.gwp = gwp;
  a.B = p->B;
  a.H = p->H;
  a.W = p->W;
  a.Cp = pl.Cp;
  a.Ho = pl.Ho;
  a.Wo = pl.Wo;
  a.Cout = p->Cout;
  a.kH = p->kH;
  a.kW = p->kW;
  a.sH = p->sH;
  a.sW = p->sW;
  a.pH = p->pH;
  a.pW = p->pW;
  a.dH = p->dH;
  a.dW = p->dW;
  a.dg = p->deformable_groups;
  a.cpg = pl.Cp / p->deformable_groups;   // dg > 1 implies Cp == Cin
  a.P = P;

  if (pl.commuted && (need_in || grad_weight)) {
    unsigned char *cw = ws + pl.xh_bytes + pl.gxh_bytes + pl.go_bytes + pl.wp_bytes + pl.gwp_bytes;
    float *yb = reinterpret_cast<float *>(cw);
    float *gyb = reinterpret_cast<float *>(cw + pl.y_bytes);
    float *gxc = reinterpret_cast<float *>(cw + pl.y_bytes + pl.gy_bytes);
    float *wqt = reinterpret_cast<float *>(cw + pl.y_bytes + pl.gy_bytes + pl.gxc_bytes);
    unsigned short *wsp = reinterpret_cast<unsigned short *>(cw + pl.y_bytes + pl.gy_bytes +
                                                             pl.gxc_bytes + pl.wqt_bytes);
    unsigned char *geo = cw + pl.y_bytes + pl.gy_bytes + pl.gxc_bytes + pl.wqt_bytes + pl.wsp_bytes;
    float *wpart = pl.wpart_bytes ? reinterpret_cast<float *>(geo + pl.geo_bytes) : nullptr;
    long long wplanes = 0;   // planes of partial sums the first chunk wrote
    const int N = T * p->Cout;
    if (need_in) {
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_weight_views_kernel<TIO>,
                                               dim3(blocks_for((long long)N * pl.Cp)), dim3(256), 0,
                                               s, static_cast<const TIO *>(weight), wp, wqt, wsp,
                                               p->Cout, p->Cin, T, pl.Cp, p->groups));
      if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
      const size_t off_elems = (size_t)p->B * 2 * T * HWo;
      if (hipMemsetAsync(grad_offset, 0, off_elems * 4, s) != hipSuccess) return GANET_ERR_HIP;
      if (grad_mask && hipMemsetAsync(grad_mask, 0, off_elems * 2, s) != hipSuccess)
        return GANET_ERR_HIP;
    }
    if (grad_weight && !wpart && hipMemsetAsync(gwp, 0, pl.gwp_bytes, s) != hipSuccess) return GANET_ERR_HIP;
    for (int img0 = 0; img0 < p->B; img0 += pl.nb) {
      const int nimg = p->B - img0 < pl.nb ? p->B - img0 : pl.nb;
      const long long M = (long long)nimg * HW;
      const float *xc = a.xh + (size_t)img0 * HW * pl.Cp;
      const float *yc = yb;
      if (need_in && y_saved) {
        yc = y_saved + (size_t)img0 * HW * N;   // kept by the forward: [B][H*W][T*Cout]
      } else if (need_in) {
        // per-tap products: three bf16 passes over high/low splits (as the forward forms them)
        const int rc2 = cts_gemm_f32_split(xc, wsp, yb, M, pl.Cp, N, s);
        if (rc2 != GANET_OK) return rc2;
      }
      // grad_Y: binned in LDS while an image has few tiles (every tile's workgroup scans the
      // whole image), through global atomics otherwise
      const int parts = (HW + BIN_Q - 1) / BIN_Q;
      int max_parts = BIN_MAX_PARTS;
      if (const char *e = getenv("GANET_BWD_BIN_MAX_PARTS")) max_parts = atoi(e);   // A/B knob
      const bool binned = parts <= max_parts && p->H < 65535 && p->W < 65535 &&
                          !getenv("GANET_BWD_SCATTER_ATOMIC");
      if (!binned && hipMemsetAsync(gyb, 0, (size_t)M * N * 4, s) != hipSuccess) return GANET_ERR_HIP;
      BwdCtsArgs c;
      c.y = need_in ? yc : nullptr;
      c.gy = gyb;
      c.go = a.go;
      c.offset = offset;
      c.mask = mask;
      c.goff = grad_offset;
      c.gmask = grad_mask;
      c.img0 = img0;
      c.nimg = nimg;
      c.H = p->H;
      c.W = p->W;
      c.Ho = pl.Ho;
      c.Wo = pl.Wo;
      c.Cout = p->Cout;
      c.T = T;
      c.kW = p->kW;
      c.pH = p->pH;
      c.pW = p->pW;
      c.dH = p->dH;
      c.dW = p->dW;
      c.geo_q = reinterpret_cast<unsigned *>(geo);
      c.geo_f = reinterpret_cast<f32x2 *>(geo + ganet_align_up((size_t)pl.nb * T * HWo * 4, 256));
      if (binned) {
        hipLaunchKernelGGL(cts_bwd_geo_kernel, dim3(blocks_for((long long)nimg * T * HWo)), dim3(256), 0, s, c);
        hipLaunchKernelGGL(cts_bwd_bin_kernel, dim3((unsigned)(nimg * T * parts), (unsigned)(p->Cout / 64)),
                           dim3(BIN_THREADS), 0, s, c, parts);
        if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
        c.gy = nullptr;
      }
      if (binned && need_in && !getenv("GANET_BWD_COORD_32LANE")) {
        long long nblk = ((long long)nimg * HWo + 15) / 16;
        if (nblk > 256 * 32) nblk = 256 * 32;
        hipLaunchKernelGGL(cts_bwd_coord_kernel, dim3((unsigned)nblk, (unsigned)(p->Cout / 64)), dim3(256),
                           0, s, c);
        if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
      } else if (need_in || !binned) {
        const long long groups = ((long long)nimg * HWo + 1) / 2;
        long long nblk = (groups + CS_WAVES - 1) / CS_WAVES;
        if (nblk > 256 * 16) nblk = 256 * 16;
        hipLaunchKernelGGL(cts_bwd_sample_kernel, dim3((unsigned)nblk, (unsigned)(p->Cout / 64)),
                           dim3(CS_WAVES * 64), 0, s, c);
        if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
      }
      if (direct_gx) {
        // grad_x += grad_Y * Wq straight into the caller's channel-last tensor (three bf16
        // passes, K-outer kernel: grad_Y is read once)
        float *gx = static_cast<float *>(grad_x) + (size_t)img0 * HW * p->Cin;
        const int rc2 = cts_gemm_f32_split_wide(gyb, wsp + 2 * (size_t)N * pl.Cp, gx, M, N, pl.Cp,
                                                p->Cin, pl.overwrite_x ? 0 : 1, s);
        if (rc2 != GANET_OK) return rc2;
      } else if (need_in) {
        const int rc2 = pl.Cp <= 256
                            ? cts_gemm_f32_split_wide(gyb, wsp + 2 * (size_t)N * pl.Cp, gxc, M, N,
                                                      pl.Cp, pl.Cp, 0, s)
                            : cts_gemm_f32(gyb, wqt, gxc, M, N, pl.Cp, s);
        if (rc2 != GANET_OK) return rc2;
        if (p->x_layout == GANET_NHWC) {
          GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(
                                    bwd_add_rows_kernel<TIO>, dim3(blocks_for(M * p->Cin)), dim3(256),
                                    0, s, gxc, static_cast<TIO *>(grad_x) + (size_t)img0 * HW * p->Cin,
                                    M, p->Cin, pl.Cp));
        } else {
          dim3 grid((HW + 63) / 64, (pl.Cp + 63) / 64, nimg);
          GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(
                                    bwd_add_nhwc_to_nchw_kernel<TIO>, grid, dim3(256), 0, s, gxc,
                                    static_cast<TIO *>(grad_x) + (size_t)img0 * p->Cin * HW, p->Cin,
                                    HW, pl.Cp));
        }
        if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
      }
      long long planes_now = 0;
      if (grad_weight && launch_wgrad_split(gyb, xc, gwp, M, N, pl.Cp, wpart, img0 == 0, &planes_now, s)) {
        if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
        if (img0 == 0) wplanes = planes_now;
        else if (wpart && planes_now > wplanes) return GANET_ERR_HIP;   // cannot happen: chunks never grow
      } else if (grad_weight) {
        const int nblocks = (N / 64) * (pl.Cp / 64);
        long long split = 1024 / nblocks + 1;
        const long long maxsplit = (M / 2 + 255) / 256;   // >= 64 pixel pairs per wavefront
        if (split > maxsplit) split = maxsplit;
        if (split < 1) split = 1;
        hipLaunchKernelGGL(cts_bwd_wgrad_kernel, dim3((unsigned)nblocks, (unsigned)split), dim3(256),
                           0, s, gyb, xc, gwp, M, N, pl.Cp);
        if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
      }
    }
    if (grad_weight && wpart) {
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(
                                bwd_reduce_unpack_weight_kernel<TIO>,
                                dim3(blocks_for((long long)N * pl.Cp)), dim3(256), 0, s, wpart,
                                (int)wplanes, static_cast<TIO *>(grad_weight), p->Cout, p->Cin, T,
                                pl.Cp, p->groups));
      if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    } else if (grad_weight) {
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(
                                bwd_unpack_add_weight_kernel<TIO>,
                                dim3(blocks_for((long long)p->Cout * p->Cin * T)), dim3(256), 0, s,
                                gwp, static_cast<TIO *>(grad_weight), p->Cout, p->Cin, T, pl.Cp,
                                p->groups));
      if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    }
  } else {
  if (need_in) {
    GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_pack_weight_kernel<TIO>,
                                             dim3(blocks_for((long long)T * p->Cout * pl.Cp)),
                                             dim3(256), 0, s, static_cast<const TIO *>(weight), wp,
                                             p->Cout, p->Cin, T, pl.Cp, p->groups));
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    if (pl.copy_x && hipMemsetAsync(gxh, 0, pl.gxh_bytes, s) != hipSuccess) return GANET_ERR_HIP;
    const size_t off_elems = (size_t)p->B * a.dg * 2 * T * HWo;
    if (hipMemsetAsync(grad_offset, 0, off_elems * 4, s) != hipSuccess) return GANET_ERR_HIP;
    if (grad_mask && hipMemsetAsync(grad_mask, 0, off_elems * 2, s) != hipSuccess)
      return GANET_ERR_HIP;
    const long long tiles = (P + 31) / 32;
    hipLaunchKernelGGL(dcn_bwd_input_kernel, dim3((unsigned)((tiles + BW_WAVES - 1) / BW_WAVES)),
                       dim3(BW_WAVES * 64), 0, s, a);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    if (pl.copy_x) {
      if (p->x_layout == GANET_NHWC) {
        GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_add_rows_kernel<TIO>,
                                                 dim3(blocks_for(npix_in * p->Cin)), dim3(256), 0, s,
                                                 gxh, static_cast<TIO *>(grad_x), npix_in, p->Cin,
                                                 pl.Cp));
      } else {
        dim3 grid((HW + 63) / 64, (pl.Cp + 63) / 64, p->B);
        GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_add_nhwc_to_nchw_kernel<TIO>, grid, dim3(256),
                                                 0, s, gxh, static_cast<TIO *>(grad_x), p->Cin, HW,
                                                 pl.Cp));
      }
      if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    }
  }
  if (grad_weight) {
    if (hipMemsetAsync(gwp, 0, pl.gwp_bytes, s) != hipSuccess) return GANET_ERR_HIP;
    const long long tiles = (P + 31) / 32;
    long long split = (tiles + BW_WAVES * 8 - 1) / (BW_WAVES * 8);   // >= 8 tiles per wave
    const long long want = 2048 / ((long long)T * (pl.Cp / 32)) + 1;
    if (split > want) split = want;
    if (split < 1) split = 1;
    dim3 grid((unsigned)(T * (pl.Cp / 32)), (unsigned)split);
    hipLaunchKernelGGL(dcn_bwd_weight_kernel, grid, dim3(BW_WAVES * 64), 0, s, a);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(
                              bwd_unpack_add_weight_kernel<TIO>,
                              dim3(blocks_for((long long)p->Cout * p->Cin * T)), dim3(256), 0, s, gwp,
                              static_cast<TIO *>(grad_weight), p->Cout, p->Cin, T, pl.Cp, p->groups));
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }
  }  // direct form
  if (grad_bias) {
    long long nb = (P + 1023) / 1024;
    if (nb > 1024) nb = 1024;
    float *gb = static_cast<float *>(grad_bias);
    if (dt != GANET_F32) {   // summed in float32 scratch, added to the 16-bit tensor afterwards
      gb = reinterpret_cast<float *>(ws + need - pl.gb_bytes);
      if (hipMemsetAsync(gb, 0, pl.gb_bytes, s) != hipSuccess) return GANET_ERR_HIP;
    }
    hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3((unsigned)nb), dim3(256), 0, s, a.go, gb, P,
                       p->Cout);
    if (dt != GANET_F32)
      GANET_IO_DISPATCH(dt, hipLaunchKernelGGL(bwd_add_rows_kernel<TIO>, dim3(1), dim3(256), 0, s, gb,
                                               static_cast<TIO *>(grad_bias), 1ll, p->Cout, p->Cout));
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }
  return GANET_OK;
}
}  // namespace

extern "C" int ganet_deform_conv_backward(const void *x, const float *offset,
                                          const float *mask, const void *weight,
                                          const void *grad_out, void *grad_x,
                                          float *grad_offset, float *grad_mask,
                                          void *grad_weight, void *grad_bias,
                                          const ganet_dcn_params *p, void *workspace,
                                          size_t workspace_bytes, ganet_stream_t stream) {
  return backward_impl(x, offset, mask, weight, grad_out, grad_x, grad_offset, grad_mask,
                       grad_weight, grad_bias, p, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int ganet_deform_conv_backward_y(const void *x, const float *offset,
                                            const float *mask, const void *weight,
                                            const void *grad_out, void *grad_x,
                                            float *grad_offset, float *grad_mask,
                                            void *grad_weight, void *grad_bias,
                                            const ganet_dcn_params *p, void *workspace,
                                            size_t workspace_bytes, ganet_stream_t stream,
                                            const void *y_saved) {
  if (!y_saved) return GANET_ERR_NULL;
  return backward_impl(x, offset, mask, weight, grad_out, grad_x, grad_offset, grad_mask,
                       grad_weight, grad_bias, p, workspace, workspace_bytes, stream,
                       static_cast<const float *>(y_saved));
}
