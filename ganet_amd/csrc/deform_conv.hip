// Deformable convolution forward for gfx950 (MI355X): the Lane-aware Feature
// Aggregator's gather-and-aggregate as ONE fused kernel - bilinear sampling of a
// channel-last feature map straight into an LDS tile, contracted on the matrix
// cores (bf16 / f16 MFMA 32x32x16 with fp32 accumulate; float32 tensors as three bf16 passes
// over hi/lo planes of the blended tile - SPLIT below - or, on request, the exact-fp32 MFMA
// 32x32x2).
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

#include "dcn_device.h"

namespace {

using namespace ganet_dcn;

// Bilinear blend of four 16-byte pieces (one per corner) with per-pixel weights,
// two channels per instruction (v_pk_mul/fma_f32).
//   w1*v1 + w2*v2 + w3*v3 + w4*v4   (deform_conv_cuda_kernel.cu:111-113)
// The DCNv2 mask (:627, `val * mask`) is folded into the four weights by the
// caller (w_i * mask): same value up to one fp32 rounding, and branch-free.
template <typename TA>
struct Blend;

template <>
struct Blend<float> {
  __device__ static __forceinline__ u32x4 run(const u32x4 *v, const float4 &w) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      f32x2 a = {__uint_as_float(v[0][2 * i]), __uint_as_float(v[0][2 * i + 1])};
      f32x2 b = {__uint_as_float(v[1][2 * i]), __uint_as_float(v[1][2 * i + 1])};
      f32x2 c = {__uint_as_float(v[2][2 * i]), __uint_as_float(v[2][2 * i + 1])};
      f32x2 d = {__uint_as_float(v[3][2 * i]), __uint_as_float(v[3][2 * i + 1])};
      f32x2 r = a * w.x;
      r = b * w.y + r;
      r = c * w.z + r;
      r = d * w.w + r;
      o[2 * i] = __float_as_uint(r.x);
      o[2 * i + 1] = __float_as_uint(r.y);
    }
    return o;
  }
};

template <>
struct Blend<bf16_t> {
  __device__ static __forceinline__ f32x2 up(unsigned int u) {
    f32x2 r = {__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
    return r;
  }
  __device__ static __forceinline__ u32x4 run(const u32x4 *v, const float4 &w) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 r = up(v[0][i]) * w.x;
      r = up(v[1][i]) * w.y + r;
      r = up(v[2][i]) * w.z + r;
      r = up(v[3][i]) * w.w + r;
      typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
      const bf16x2 p = {(__bf16)r.x, (__bf16)r.y};  // v_cvt_pk_bf16_f32: RNE, NaN kept
      o[i] = __builtin_bit_cast(unsigned int, p);
    }
    return o;
  }
};

template <>
struct Blend<f16_t> {
  typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
  __device__ static __forceinline__ f32x2 up(unsigned int u) {
    const f16x2 h = __builtin_bit_cast(f16x2, u);
    f32x2 r = {(float)h.x, (float)h.y};
    return r;
  }
  __device__ static __forceinline__ u32x4 run(const u32x4 *v, const float4 &w) {
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x2 r = up(v[0][i]) * w.x;
      r = up(v[1][i]) * w.y + r;
      r = up(v[2][i]) * w.z + r;
      r = up(v[3][i]) * w.w + r;
      const f16x2 p = {(_Float16)r.x, (_Float16)r.y};
      o[i] = __builtin_bit_cast(unsigned int, p);
    }
    return o;
  }
};

// Compiler-level ordering point for LDS traffic that stays inside one wavefront
// (the hardware executes a wave's LDS operations in order).
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One workgroup = NW wavefronts; every wavefront owns 32 output pixels x 64 output
// channels end to end (coordinates, gather, blend, its private A tile in LDS, two
// 32x32 accumulators).  Only the weight chunk is shared: it is double-buffered in
// LDS and changes once per tap (bf16/f16: 256 channels) or half tap (fp32: 128),
// which is the only workgroup barrier in the loop.  The gather is software
// pipelined one step (16 buffer loads per lane-group) ahead of the blend, across
// chunk and tap boundaries, so the memory pipe stays busy while a wave blends,
// writes LDS and issues its MFMAs.
//
// SPLIT (float32 tensors, ganet_dcn_params.mfma_bf16 == 0): the gather and the blend stay
// float32, but the blended tile goes to LDS as two bf16 planes per row - hi = bf16(v),
// lo = bf16(v - hi) - the weights arrive pre-split the same way (pack_weight_split_kernel),
// and the contraction runs on the bf16 matrix cores as hi*hi + hi*lo + lo*hi with float32
// accumulation (products of two bf16 are exact in float32; what is dropped is <= 3 * 2^-18 of
// a term): 12 MFMA 32x32x16 per 32-channel chunk instead of 32 float32 MFMA 32x32x2 at 1/16 of
// the rate.  Same LDS bytes as the float32 tiles.
template <typename TA, typename TO, int KCB, int NW, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64, 2) void dcn_fused_kernel(DcnArgs a) {
  static_assert(!SPLIT || sizeof(TA) == 4, "the split form contracts float32 tensors");
  using TM = typename std::conditional<SPLIT, bf16_t, TA>::type;   // LDS tile / MFMA operand element
  constexpr int ES = sizeof(TA);
  constexpr int MS = sizeof(TM);
  constexpr int EPL = 16 / ES;                          // gathered elements per 16-B piece
  constexpr int EPM = 16 / MS;                          // tile elements per 16-B piece
  constexpr int KC = (KCB * ES > 128) ? 128 / ES : KCB; // gather chunk (elements)
  constexpr int NSUB = KCB / KC;                        // gather chunks per weight chunk
  constexpr int LPP = KC / EPL;                         // lanes per pixel chunk
  constexpr int PPL = 64 / LPP;                         // pixels per load instruction
  constexpr int NPASS = 32 / PPL;                       // load passes per gather chunk
  constexpr int GROUP = NPASS;                          // one pipeline step = one chunk
  constexpr int PLANES = SPLIT ? 2 : 1;                 // hi | lo side by side in a tile row
  constexpr int LDA = PLANES * KC + EPM;                // +16 B: conflict-free b128 reads
  constexpr int LDB = PLANES * KCB + EPM;
  constexpr int TMW = 32;                               // pixels per wavefront
  constexpr int TMB = NW * TMW;                         // pixels per workgroup
  typedef typename Mma<TM>::frag frag_t;

  constexpr int B_BYTES = TN * LDB * MS;                // one weight buffer
  constexpr int A_BYTES = TMW * LDA * MS;               // one wave's A tile (x2: double buffer)
  constexpr int C_BYTES = 32 * (TMW + 1) * 4;           // NCHW transpose of one 32x32 block
  constexpr int A0_BYTES = A_BYTES > C_BYTES ? A_BYTES : C_BYTES;  // A buffer 0 / transpose
  __shared__ __attribute__((aligned(16))) unsigned char s_b[2 * B_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_a0[NW * A0_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_a1[NW * A_BYTES];
  // per wave, double buffered over (tap, deformable group): byte offsets of the
  // four bilinear corners inside this tile's buffer window (OOB_OFFSET = corner
  // outside the map: the buffer load returns 0, the reference's `v = 0`) and the
  // four bilinear weights (times the DCNv2 mask value)
  __shared__ __attribute__((aligned(16))) unsigned int s_off[NW][2][TMW][4];
  __shared__ __attribute__((aligned(16))) float s_w[NW][2][TMW][4];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware tile order (blocks b and b+8 share an L2): each XCD walks one
  // contiguous range of tiles, so re-reads of an image stay in one L2.
  // Bijective for any grid size; placement only affects speed.
  long long tile;
  {
    const unsigned nwg = gridDim.x, bid = blockIdx.x;
    const unsigned q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    tile = (long long)(xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
  }
  const int n0 = blockIdx.y * TN;
  const int HWo = a.Ho * a.Wo;
  const int T = a.kH * a.kW;
  const TM *wp = static_cast<const TM *>(a.wp);
  const size_t w_plane = (size_t)T * a.CoutP * a.Cp;          // SPLIT: the lo plane follows the hi plane
  TM *A0 = reinterpret_cast<TM *>(s_a0 + wid * A0_BYTES);   // chunks 0, 2, 4, ...
  TM *A1 = reinterpret_cast<TM *>(s_a1 + wid * A_BYTES);    // chunks 1, 3, 5, ...

  // Buffer window of this tile: starts at the first image the tile touches and is
  // clamped below 2 GiB, so 32-bit offsets address it and OOB_OFFSET is outside.
  const int b0 = (int)((tile * TMB) / HWo);
  const size_t img_bytes = (size_t)a.H * a.W * a.Cp * ES;
  const size_t win_rest = (size_t)(a.B - b0) * img_bytes;
  const unsigned win_bytes = win_rest < 0x7fffffffull ? (unsigned)win_rest : 0x7fffffffu;
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char *>(static_cast<const unsigned char *>(a.xh)) +
          (size_t)b0 * img_bytes,
      0, (int)win_bytes, 0x00020000);

  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    acc0[i] = 0.f;
    acc1[i] = 0.f;
  }

  const int lp = lane % LPP;   // 16-B piece within the pixel chunk
  const int pq = lane / LPP;   // pixel within one load instruction
  const bool has_mask = a.mask != nullptr;

  // this lane's output pixel for the coordinate computation (lanes 0..31)
  const long long my_p = tile * TMB + wid * TMW + (lane & 31);
  const bool my_valid = my_p < a.P;
  const int my_b = my_valid ? (int)(my_p / HWo) : 0;
  const int my_hw = my_valid ? (int)(my_p - (long long)my_b * HWo) : 0;
  const int my_ho = my_hw / a.Wo, my_wo = my_hw - my_ho * a.Wo;

  // offsets (and mask) of (tap t, group g) for this lane's pixel: global loads
  auto load_offsets = [&](int t, int g, float &oh, float &ow, float &mk) {
    oh = 0.f;
    ow = 0.f;
    mk = 1.f;
    if (my_valid && lane < 32) {
      const float *op = a.offset + ((size_t)(my_b * a.dg + g) * 2 * T + 2 * t) * HWo + my_hw;
      oh = ldg_f32(op);
      ow = ldg_f32(op + HWo);
      if (has_mask)
        mk = ldg_f32(a.mask + ((size_t)(my_b * a.dg + g) * T + t) * HWo + my_hw);
    }
  };
  // sampling position, corner offsets and bilinear weights -> table `slot`
  auto store_coords = [&](int t, int slot, float off_h, float off_w, float mk) {
    if (lane < 32) {
      const int ti = t / a.kW, tj = t - ti * a.kW;
      unsigned c1 = OOB_OFFSET, c2 = OOB_OFFSET, c3 = OOB_OFFSET, c4 = OOB_OFFSET;
      float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
      if (my_valid) {
        if (a.points)  // dcn_offset = pts - base  (lanepoints_conv.py:63-74,170)
          off_w = off_w - (float)(tj - a.pW);
        const float h_im = (float)(my_ho * a.sH - a.pH + ti * a.dH) + off_h;
        const float w_im = (float)(my_wo * a.sW - a.pW + tj * a.dW) + off_w;
        if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
          const float hh = 1.f - lh, hw_ = 1.f - lw;
          w1 = hh * hw_;
          w2 = hh * lw;
          w3 = lh * hw_;
          w4 = lh * lw;
          const unsigned px = (unsigned)(a.Cp * ES);  // bytes per pixel
          const unsigned row = (unsigned)a.W * px;
          const unsigned base = (unsigned)(my_b - b0) * (unsigned)img_bytes +
                                (unsigned)h_low * row + (unsigned)w_low * px;
          const bool hl = h_low >= 0, hh_ok = h_high <= a.H - 1;
          const bool wl = w_low >= 0, wh_ok = w_high <= a.W - 1;
          if (hl && wl) c1 = base;
          if (hl && wh_ok) c2 = base + px;
          if (hh_ok && wl) c3 = base + row;
          if (hh_ok && wh_ok) c4 = base + row + px;
        }
      }
      *reinterpret_cast<uint4 *>(s_off[wid][slot][lane]) = make_uint4(c1, c2, c3, c4);
      *reinterpret_cast<float4 *>(s_w[wid][slot][lane]) =
          make_float4(w1 * mk, w2 * mk, w3 * mk, w4 * mk);
    }
  };

  // weight chunk: global -> registers (early) -> LDS buffer (late)
  // (SPLIT: a row's chunk is KCB/8 pieces of the hi plane, then KCB/8 of the lo plane -
  // as many 16-B pieces as the float32 chunk has)
  constexpr int PPR = PLANES * KCB / EPM;               // 16-B pieces per row and chunk
  constexpr int PPP = KCB / EPM;                        // ... per plane
  constexpr int BPIECES = TN * PPR;
  constexpr int BPT = (BPIECES + NW * 64 - 1) / (NW * 64);
  u32x4 breg[BPT];
  auto load_b = [&](int t, int cbase) {
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NW * 64;
      if (BPIECES % (NW * 64) == 0 || q < BPIECES) {
        const int row = q / PPR, pc = q - row * PPR;
        const int plane = pc / PPP, pw = pc - plane * PPP;
        breg[j] = ldg128(wp + plane * w_plane +
                         ((size_t)(t * a.CoutP + n0 + row) * a.Cp + cbase + pw * EPM));
      }
    }
  };
  auto store_b = [&](int buf) {
    TM *Bs = reinterpret_cast<TM *>(s_b + buf * B_BYTES);
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NW * 64;
      if (BPIECES % (NW * 64) == 0 || q < BPIECES) {
        const int row = q / PPR, pc = q - row * PPR;
        const int plane = pc / PPP, pw = pc - plane * PPP;
        *reinterpret_cast<u32x4 *>(Bs + row * LDB + plane * KCB + pw * EPM) = breg[j];
      }
    }
  };

  u32x4 va[GROUP][4], vb[GROUP][4];
  auto issue = [&](u32x4 (&v)[GROUP][4], int slot, int cabs) {
    const unsigned lane_off = (unsigned)((cabs + lp * EPL) * ES);
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int r = u * PPL + pq;
      const uint4 cx = *reinterpret_cast<const uint4 *>(s_off[wid][slot][r]);
#ifdef GANET_ABLATE  // profiling build only (tools/ablate.sh): skip the gather loads
      if (a.ablate & 2) {
        v[u][0] = v[u][1] = v[u][2] = v[u][3] = u32x4{cx.x, cx.y, cx.z, cx.w + lane_off};
        continue;
      }
#endif
      v[u][0] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.x + lane_off, 0, 0);
      v[u][1] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.y + lane_off, 0, 0);
      v[u][2] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.z + lane_off, 0, 0);
      v[u][3] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.w + lane_off, 0, 0);
    }
  };
  auto blend = [&](u32x4 (&v)[GROUP][4], int slot, TM *Adst) {
#pragma unroll
    for (int u = 0; u < GROUP; ++u) {
      const int r = u * PPL + pq;
      const float4 w = *reinterpret_cast<const float4 *>(s_w[wid][slot][r]);
      const u32x4 o = Blend<TA>::run(v[u], w);
      if constexpr (SPLIT) {
        // four blended floats -> 4 hi + 4 lo bf16, 8 bytes each
        uint2 hi, lo;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float x0 = __uint_as_float(o[2 * k]), x1 = __uint_as_float(o[2 * k + 1]);
          // hi = x cut to its upper 16 bits (exact), lo = x - hi rounded to nearest even on
          // the 16 bits it drops: integer operations only.  (v_cvt_pk_bf16_f32 is NOT used here:
          // with this instruction in the blend, pixels held by lanes 48-63 came out wrong now and
          // then whenever two wavefronts shared a SIMD - 12 x 34x106 x Cin 64 and every other
          // shape at occupancy 2, never at occupancy 1; waits, no-ops and full barriers around it
          // changed nothing, these integer forms made it go away.  Recorded in DESIGN 3.1 (1s).)
          const unsigned int u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
          const unsigned int h0 = u0 & 0xffff0000u, h1 = u1 & 0xffff0000u;
          const unsigned int l0 = __float_as_uint(x0 - __uint_as_float(h0));
          const unsigned int l1 = __float_as_uint(x1 - __uint_as_float(h1));
          const unsigned int r0 = l0 + 0x7fffu + ((l0 >> 16) & 1u);
          const unsigned int r1 = l1 + 0x7fffu + ((l1 >> 16) & 1u);
          (k == 0 ? hi.x : hi.y) = (h0 >> 16) | h1;
          (k == 0 ? lo.x : lo.y) = (r0 >> 16) | (r1 & 0xffff0000u);
        }
        *reinterpret_cast<uint2 *>(Adst + r * LDA + lp * EPL) = hi;
        *reinterpret_cast<uint2 *>(Adst + r * LDA + KC + lp * EPL) = lo;
      } else {
        *reinterpret_cast<u32x4 *>(Adst + r * LDA + lp * EPL) = o;
      }
    }
  };

  const int CPGK = a.cpg / KC;          // gather chunks per (tap, group)
  const int NTG = T * a.dg;             // (tap, group) pairs
  const int NA = NTG * CPGK;            // gather chunks in total

  // Chunk j = (tg, c): tg = j / CPGK enumerates (tap, group) pairs tap-major.
  // Three cursors run through the chunks: MFMA at k, blend at k+1, loads at k+2.
  float noh, now_, nmk;                 // prefetched offsets of the next (tap, group)
  int tg_coords = 0;                    // (tap, group) pairs whose table is written
  // make the coordinate table of pair `tg` available (tables are double buffered)
  auto ensure_coords = [&](int tg) {
    if (tg == tg_coords) {
      store_coords(tg / a.dg, tg & 1, noh, now_, nmk);
      wave_lds_sync();
      ++tg_coords;
      if (tg_coords < NTG) load_offsets(tg_coords / a.dg, tg_coords % a.dg, noh, now_, nmk);
    }
  };
  auto issue_chunk = [&](u32x4 (&v)[GROUP][4], int j) {
    const int tg = j / CPGK, c = j - tg * CPGK;
    ensure_coords(tg);
    issue(v, tg & 1, (tg % a.dg) * a.cpg + c * KC);
  };
  auto wchunk_origin = [&](int nb, int &t, int &cbase) {   // weight chunk nb
    const int j = nb * NSUB;
    const int tg = j / CPGK, c = j - tg * CPGK;
    t = tg / a.dg;
    cbase = (tg % a.dg) * a.cpg + c * KC;
  };

  // ---- prologue ---------------------------------------------------------------
  load_offsets(0, 0, noh, now_, nmk);
  load_b(0, 0);
  issue_chunk(va, 0);
  store_b(0);
  if (NA > 1) issue_chunk(vb, 1);
  blend(va, 0, A0);

  // One step: loads of chunk k+2 go out, chunk k+1 is blended into the other A
  // buffer, chunk k is contracted - three independent instruction streams (VMEM
  // issue, VALU + ds_write, ds_read + MFMA) that sit in ONE basic block so the
  // scheduler can interleave them; everything conditional (barrier, weight
  // staging, coordinate tables) happens before or after that block.
  auto step = [&](int k, u32x4 (&vcur)[GROUP][4], u32x4 (&vnxt)[GROUP][4],
                  const TM *Acur, TM *Anxt, auto do_issue, auto do_blend) {
    const int sub = k % NSUB;
    const int nbi = k / NSUB;           // weight chunk index
    if (sub == 0) {
      // weight chunk nbi was written to its buffer by every wave: one barrier
      __syncthreads();
      if ((nbi + 1) * NSUB < NA) {      // prefetch the next weight chunk into registers
        int t3, cb3;
        wchunk_origin(nbi + 1, t3, cb3);
        load_b(t3, cb3);
      }
    } else {
      wave_lds_sync();                  // A[k&1] written in the previous step
    }
    int islot = 0, icabs = 0;
    if (decltype(do_issue)::value) {
      const int tg = (k + 2) / CPGK, c = (k + 2) - tg * CPGK;
      ensure_coords(tg);
      islot = tg & 1;
      icabs = (tg % a.dg) * a.cpg + c * KC;
    }
    const int bslot = ((k + 1) / CPGK) & 1;
    // ---- hot block -----------------------------------------------------------
    // program order: loads, MFMA operands + MFMAs, blend.  The blend's VALU work
    // does not depend on the MFMAs and its LDS writes go to the other A buffer,
    // so the scheduler is free to pull it up between the MFMAs.
    if (decltype(do_issue)::value) issue(vcur, islot, icabs);
    {
      const int i = lane & 31, h = lane >> 5;
      const TM *Bs = reinterpret_cast<const TM *>(s_b + (nbi & 1) * B_BYTES);
      const TM *arow = Acur + i * LDA + h * EPM;
      const TM *brow0 = Bs + i * LDB + sub * KC + h * EPM;
      const TM *brow1 = brow0 + 32 * LDB;
      if constexpr (SPLIT) {
#pragma unroll
        for (int q = 0; q < KC / (2 * EPM); ++q) {
          const frag_t ah = *reinterpret_cast<const frag_t *>(arow + q * 2 * EPM);
          const frag_t al = *reinterpret_cast<const frag_t *>(arow + KC + q * 2 * EPM);
          const frag_t bh0 = *reinterpret_cast<const frag_t *>(brow0 + q * 2 * EPM);
          const frag_t bl0 = *reinterpret_cast<const frag_t *>(brow0 + KCB + q * 2 * EPM);
          const frag_t bh1 = *reinterpret_cast<const frag_t *>(brow1 + q * 2 * EPM);
          const frag_t bl1 = *reinterpret_cast<const frag_t *>(brow1 + KCB + q * 2 * EPM);
          // small terms first, as the float32 sum would want them
          Mma<TM>::mma(al, bh0, acc0);
          Mma<TM>::mma(al, bh1, acc1);
          Mma<TM>::mma(ah, bl0, acc0);
          Mma<TM>::mma(ah, bl1, acc1);
          Mma<TM>::mma(ah, bh0, acc0);
          Mma<TM>::mma(ah, bh1, acc1);
        }
      } else
#pragma unroll
      for (int q = 0; q < KC / (2 * EPL); ++q) {
        const frag_t fa = *reinterpret_cast<const frag_t *>(arow + q * 2 * EPL);
        const frag_t fb0 = *reinterpret_cast<const frag_t *>(brow0 + q * 2 * EPL);
        const frag_t fb1 = *reinterpret_cast<const frag_t *>(brow1 + q * 2 * EPL);
#ifdef GANET_ABLATE  // profiling build only: skip the MFMAs, keep their operands live
        if (a.ablate & 1) {
          asm volatile("" ::"v"(fa), "v"(fb0), "v"(fb1));
          continue;
        }
#endif
        Mma<TA>::mma(fa, fb0, acc0);
        Mma<TA>::mma(fa, fb1, acc1);
      }
    }
    if (decltype(do_blend)::value) blend(vnxt, bslot, Anxt);
    // Interleave hint for the in-order issue: after each MFMA a few VALU (blend /
    // address), one LDS and (every other slot) one buffer-load instruction, so the
    // matrix pipe runs while this wave's vector and memory work issues.
    {
      constexpr int NMFMA = SPLIT ? (KC / (2 * EPM)) * 6 : (KC / (2 * EPL)) * 2 * (ES == 4 ? 4 : 1);
      constexpr int VPER = SPLIT ? 8 : (ES == 4 ? 2 : 10);
#pragma unroll
      for (int i = 0; i < NMFMA; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, VPER, 0);   // VALU
        __builtin_amdgcn_sched_group_barrier(0x080, SPLIT ? 3 : 1, 0);      // DS
        if ((i & 1) == 0 || ES != 4 || SPLIT)
          __builtin_amdgcn_sched_group_barrier(0x020, (ES == 4 && !SPLIT) ? 1 : 2, 0);  // VMEM read
      }
    }
    // ---------------------------------------------------------------------------
    if (sub == NSUB - 1 && k + 1 < NA) {
      // this wave is done with weight chunk nbi; the other buffer was last read
      // in chunk nbi-1, which every wave left before the barrier above
      store_b((nbi + 1) & 1);
    }
  };
  {
    const std::true_type yes{};
    const std::false_type no{};
    int k = 0;
    for (; k + 3 < NA; k += 2) {
      step(k, va, vb, A0, A1, yes, yes);
      step(k + 1, vb, va, A1, A0, yes, yes);
    }
    const int rest = NA - k;            // 1, 2 or 3 chunks left (k is even)
    if (rest == 3) {
      step(k, va, vb, A0, A1, yes, yes);
      step(k + 1, vb, va, A1, A0, no, yes);
      step(k + 2, va, vb, A0, A1, no, no);
    } else if (rest == 2) {
      step(k, va, vb, A0, A1, no, yes);
      step(k + 1, vb, va, A1, A0, no, no);
    } else {
      step(k, va, vb, A0, A1, no, no);
    }
  }
  wave_lds_sync();

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
  const long long pw = tile * TMB + wid * TMW;  // first pixel of this wave
  if (a.out_nhwc) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
      const long long p = pw + row;
      if (p < a.P) {
        float x0 = acc0[r] + bias0, x1 = acc1[r] + bias1;
        if (a.relu) {
          x0 = fmaxf(x0, 0.f);
          x1 = fmaxf(x1, 0.f);
        }
        if (n0 + col < a.Cout) out[p * a.Cout + n0 + col] = from_f32<TO>(x0);
        if (n0 + 32 + col < a.Cout) out[p * a.Cout + n0 + 32 + col] = from_f32<TO>(x1);
      }
    }
  } else {
    // NCHW: transpose each 32x32 accumulator through the wave's own LDS region so
    // that each store instruction writes runs of consecutive pixels
    float *Cs = reinterpret_cast<float *>(s_a0 + wid * A0_BYTES);  // [32][TMW+1]
    const int r = lane & 31;
    const long long p = pw + r;
    const int b = (int)((p < a.P ? p : 0) / HWo);
    const int hw = (int)((p < a.P ? p : 0) - (long long)b * HWo);
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) {
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * half;
        Cs[col * (TMW + 1) + row] = nb == 0 ? acc0[rr] + bias0 : acc1[rr] + bias1;
      }
      wave_lds_sync();
#pragma unroll 4
      for (int o = half; o < 32; o += 2) {
        const int oc = n0 + nb * 32 + o;
        if (p < a.P && oc < a.Cout) {
          float x = Cs[o * (TMW + 1) + r];
          if (a.relu) x = fmaxf(x, 0.f);
          out[((size_t)b * a.Cout + oc) * HWo + hw] = from_f32<TO>(x);
        }
      }
      wave_lds_sync();
    }
  }
}

// ---- small batches: the taps of a pixel tile on different wavefronts ----------------------
// At batch 1 (BASELINE's C1: one 40x100 map, 2 MB of x) nothing is bound by bytes: the
// contract-then-sample pair is two launches of one memory round trip after another (8.7 +
// 6.2 us and the gap between them), and dcn_fused_kernel walks a pixel tile's taps and
// channel chunks one round trip after the other on one wavefront (41 us).  Here a workgroup
// owns 16 output pixels x 64 outputs and its wavefronts split the TAPS: wave w samples and
// contracts taps w, w + NWV, ... of the same 16 pixels (NWV = min(taps, 8)), every wave with
// its own coordinates, gathers, blended tile and weight fragments.  The chain of dependent
// round trips is offsets -> gathers -> store whatever the number of taps or channels: the
// weight fragments (read straight from the packed weights in MFMA 16x16x32 layout, no
// staging, no workgroup barrier in the loop) are requested right behind the offsets, and the
// gathers of up to four 128-byte channel chunks (all 256 channels of a 16-bit map) are in
// flight at once.  The partial 16 x 64 blocks meet in LDS and are summed in wave order by
// the whole workgroup (a fixed order: results are bit-identical from run to run), which also
// gives coalesced stores for both output layouts.  One deformable group; float32 tensors in
// the hi/lo three-pass form only.
constexpr int TAPS_WAVES = 8;
constexpr int TAPS_PIX = 16;
#ifdef GANET_TAPS_STAMPS
// wall-clock stamps (10 ns) of the first and last wave of the first and last workgroup:
// start, coordinates in the table, gathers issued, chunk 0 contracted (one per DEPTH chunks),
// block parked, barrier passed, outputs stored (tools/taps_stamps.py)
__device__ long long g_taps_stamps[4][12];
#endif

template <typename TM>
struct Mma16;
template <>
struct Mma16<bf16_t> {
  __device__ static __forceinline__ void mma(const bf16x8 &a, const bf16x8 &b, f32x4 &c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma16<f16_t> {
  __device__ static __forceinline__ void mma(const f16x8 &a, const f16x8 &b, f32x4 &c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

template <typename TA, typename TO, bool SPLIT>
__global__ __launch_bounds__(TAPS_WAVES * 64) void dcn_taps_kernel(DcnArgs a) {
  static_assert(SPLIT == (sizeof(TA) == 4), "float32 tensors take the split form, 16-bit ones the plain one");
  using TM = typename std::conditional<SPLIT, bf16_t, TA>::type;
  constexpr int ES = sizeof(TA);
  constexpr int EPL = 16 / ES;
  constexpr int KC = 128 / ES;                          // gather chunk: 128 B of channels per pixel
  constexpr int LPP = 8;                                // lanes per pixel chunk
  constexpr int PPL = 8;                                // pixels per load instruction
  constexpr int NPASS = TAPS_PIX / PPL;                 // load passes per chunk
  constexpr int DEPTH = 4;                              // chunks whose gathers are in flight
  constexpr int PLANES = SPLIT ? 2 : 1;
  constexpr int LDA = PLANES * KC + 8;                  // 16-bit elements; +16 B: conflict-free b128 reads
  constexpr int NKS = KC / 32;                          // MFMA K steps (32 channels) per chunk
  constexpr int A_BYTES = TAPS_PIX * LDA * 2;
  constexpr int RROW = 65;                              // floats per pixel row of a partial block
  constexpr int R_BYTES = TAPS_PIX * RROW * 4;
  constexpr int W_BYTES = (2 * A_BYTES > R_BYTES ? 2 * A_BYTES : R_BYTES);   // per wave: two tiles / one block
  typedef typename Mma<TM>::frag frag_t;
  __shared__ __attribute__((aligned(16))) unsigned char s_t[TAPS_WAVES * W_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned int s_off[TAPS_WAVES][TAPS_PIX][4];
  __shared__ __attribute__((aligned(16))) float s_w[TAPS_WAVES][TAPS_PIX][4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nwv = (int)(blockDim.x >> 6);
  const long long tile = blockIdx.x;
  const int n0 = blockIdx.y * TN;
  const int HWo = a.Ho * a.Wo;
  const int T = a.kH * a.kW;
  const TM *wp = static_cast<const TM *>(a.wp_frag);   // the fragment-major copy (frag_major_index)
  const size_t w_plane = (size_t)T * a.CoutP * a.Cp;
  TM *A0 = reinterpret_cast<TM *>(s_t + wid * W_BYTES);
  TM *A1 = reinterpret_cast<TM *>(s_t + wid * W_BYTES + A_BYTES);

  // buffer window as in dcn_fused_kernel: from the first image the tile touches, below 2 GiB
  const int b0 = (int)((tile * TAPS_PIX) / HWo);
  const size_t img_bytes = (size_t)a.H * a.W * a.Cp * ES;
  const size_t win_rest = (size_t)(a.B - b0) * img_bytes;
  const unsigned win_bytes = win_rest < 0x7fffffffull ? (unsigned)win_rest : 0x7fffffffu;
  const __amdgpu_buffer_rsrc_t xsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char *>(static_cast<const unsigned char *>(a.xh)) + (size_t)b0 * img_bytes,
      0, (int)win_bytes, 0x00020000);

  const int lp = lane % LPP, pq = lane / LPP;
  const int i = lane & 15, kg = lane >> 4;            // MFMA 16x16x32: row / column i, K group kg
  const long long my_p = tile * TAPS_PIX + i;
  const bool my_valid = my_p < a.P;
  const int my_b = my_valid ? (int)(my_p / HWo) : 0;
  const int my_hw = my_valid ? (int)(my_p - (long long)my_b * HWo) : 0;
  const int my_ho = my_hw / a.Wo, my_wo = my_hw - my_ho * a.Wo;

  f32x4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

  // sampling position of (pixel i, tap t): corner offsets inside the window and the four
  // bilinear weights (times the DCNv2 mask) -> this wave's table
  // (deform_conv_cuda_kernel.cu:84-115, :227-229; lanepoints_conv.py:63-74,170 for `points`)
  auto load_offsets = [&](int t, float &off_h, float &off_w, float &mk) {
    off_h = 0.f;
    off_w = 0.f;
    mk = 1.f;
    if (lane < TAPS_PIX && my_valid) {
      const float *op = a.offset + ((size_t)my_b * 2 * T + 2 * t) * HWo + my_hw;
      off_h = ldg_f32(op);
      off_w = ldg_f32(op + HWo);
      if (a.mask) mk = ldg_f32(a.mask + ((size_t)my_b * T + t) * HWo + my_hw);
    }
  };
  auto coords = [&](int t, float off_h, float off_w, float mk) {
    if (lane < TAPS_PIX) {
      const int ti = t / a.kW, tj = t - ti * a.kW;
      unsigned c1 = OOB_OFFSET, c2 = OOB_OFFSET, c3 = OOB_OFFSET, c4 = OOB_OFFSET;
      float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
      if (my_valid) {
        if (a.points) off_w = off_w - (float)(tj - a.pW);
        const float h_im = (float)(my_ho * a.sH - a.pH + ti * a.dH) + off_h;
        const float w_im = (float)(my_wo * a.sW - a.pW + tj * a.dW) + off_w;
        if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
          const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
          const int h_high = h_low + 1, w_high = w_low + 1;
          const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
          const float hh = 1.f - lh, hw_ = 1.f - lw;
          w1 = hh * hw_;
          w2 = hh * lw;
          w3 = lh * hw_;
          w4 = lh * lw;
          const unsigned px = (unsigned)(a.Cp * ES);
          const unsigned row = (unsigned)a.W * px;
          const unsigned base = (unsigned)(my_b - b0) * (unsigned)img_bytes +
                                (unsigned)h_low * row + (unsigned)w_low * px;
          const bool hl = h_low >= 0, hh_ok = h_high <= a.H - 1;
          const bool wl = w_low >= 0, wh_ok = w_high <= a.W - 1;
          if (hl && wl) c1 = base;
          if (hl && wh_ok) c2 = base + px;
          if (hh_ok && wl) c3 = base + row;
          if (hh_ok && wh_ok) c4 = base + row + px;
        }
      }
      *reinterpret_cast<uint4 *>(s_off[wid][lane]) = make_uint4(c1, c2, c3, c4);
      *reinterpret_cast<float4 *>(s_w[wid][lane]) = make_float4(w1 * mk, w2 * mk, w3 * mk, w4 * mk);
    }
  };
  auto issue = [&](u32x4 (&v)[NPASS][4], int cabs) {
    const unsigned lane_off = (unsigned)((cabs + lp * EPL) * ES);
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      const uint4 cx = *reinterpret_cast<const uint4 *>(s_off[wid][u * PPL + pq]);
      v[u][0] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.x + lane_off, 0, 0);
      v[u][1] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.y + lane_off, 0, 0);
      v[u][2] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.z + lane_off, 0, 0);
      v[u][3] = __builtin_amdgcn_raw_buffer_load_b128(xsrc, cx.w + lane_off, 0, 0);
    }
  };
  auto blend = [&](u32x4 (&v)[NPASS][4], TM *Adst) {
#pragma unroll
    for (int u = 0; u < NPASS; ++u) {
      const int r = u * PPL + pq;
      const float4 w = *reinterpret_cast<const float4 *>(s_w[wid][r]);
      const u32x4 o = Blend<TA>::run(v[u], w);
      if constexpr (SPLIT) {
        // hi = the upper 16 bits, lo = the rest rounded to nearest even, integer operations
        // only (see the note in dcn_fused_kernel's blend)
        uint2 hi, lo;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const float x0 = __uint_as_float(o[2 * k]), x1 = __uint_as_float(o[2 * k + 1]);
          const unsigned int u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
          const unsigned int h0 = u0 & 0xffff0000u, h1 = u1 & 0xffff0000u;
          const unsigned int l0 = __float_as_uint(x0 - __uint_as_float(h0));
          const unsigned int l1 = __float_as_uint(x1 - __uint_as_float(h1));
          const unsigned int r0 = l0 + 0x7fffu + ((l0 >> 16) & 1u);
          const unsigned int r1 = l1 + 0x7fffu + ((l1 >> 16) & 1u);
          (k == 0 ? hi.x : hi.y) = (h0 >> 16) | h1;
          (k == 0 ? lo.x : lo.y) = (r0 >> 16) | (r1 & 0xffff0000u);
        }
        *reinterpret_cast<uint2 *>(Adst + r * LDA + lp * EPL) = hi;
        *reinterpret_cast<uint2 *>(Adst + r * LDA + KC + lp * EPL) = lo;
      } else {
        *reinterpret_cast<u32x4 *>(Adst + r * LDA + lp * EPL) = o;
      }
    }
  };
  // weight fragments of (tap t, channels cabs ..): lane (i, kg) supplies row n0 + 16 nb + i,
  // channels 32 ks + 8 kg .. + 8, 1 KB in a row per instruction from the fragment-major copy
  auto load_w = [&](frag_t (&wf)[PLANES][NKS][4], int t, int cabs) {
    const size_t blk = ((size_t)t * (a.Cp / 32) + cabs / 32) * (a.CoutP / 16) + n0 / 16;
#pragma unroll
    for (int pn = 0; pn < PLANES; ++pn)
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
          wf[pn][ks][nb] = __builtin_bit_cast(
              frag_t, ldg128(wp + pn * w_plane +
                             ((blk + (size_t)ks * (a.CoutP / 16) + nb) * 64 + lane) * 8));
  };
  auto contract = [&](const TM *Acur, const frag_t (&wf)[PLANES][NKS][4]) {
    const TM *arow = Acur + i * LDA + kg * 8;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
      const frag_t ah = *reinterpret_cast<const frag_t *>(arow + ks * 32);
      if constexpr (SPLIT) {
        const frag_t al = *reinterpret_cast<const frag_t *>(arow + KC + ks * 32);
        // small terms first, as in dcn_fused_kernel
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) Mma16<TM>::mma(al, wf[0][ks][nb], acc[nb]);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) Mma16<TM>::mma(ah, wf[PLANES - 1][ks][nb], acc[nb]);
      }
#pragma unroll
      for (int nb = 0; nb < 4; ++nb) Mma16<TM>::mma(ah, wf[0][ks][nb], acc[nb]);
    }
  };

#ifdef GANET_TAPS_STAMPS   // profiling build (GANET_TAPS_DEFS=-DGANET_TAPS_STAMPS): phase times of four waves
  const int stamp_b = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : -1);
  const int stamp_w = wid == 0 ? 0 : (wid == nwv - 1 ? 1 : -1);
  const int stamp_slot = (stamp_b < 0 || stamp_w < 0) ? -1 : stamp_b + stamp_w;
  int nts = 0;
#define TAPS_STAMP() do { if (stamp_slot >= 0 && lane == 0 && nts < 12) g_taps_stamps[stamp_slot][nts++] = wall_clock64(); } while (0)
#else
#define TAPS_STAMP() do { } while (0)
#endif
  TAPS_STAMP();
  const int NCH = a.Cp / KC;
  u32x4 v0[NPASS][4], v1[NPASS][4], v2[NPASS][4], v3[NPASS][4];
  frag_t wa[PLANES][NKS][4], wb[PLANES][NKS][4];
  // chunk c of tap t: its gathers sit in `v` (issued DEPTH chunks ago), its weights in `wcur`;
  // the register sets it frees are refilled with chunk c + DEPTH's gathers and chunk c + 2's weights
  auto step = [&](int t, int c, u32x4 (&v)[NPASS][4], TM *Adst, frag_t (&wcur)[PLANES][NKS][4]) {
    blend(v, Adst);
    if (c + DEPTH < NCH) issue(v, (c + DEPTH) * KC);
    wave_lds_sync();
    contract(Adst, wcur);
    if (c + 2 < NCH) load_w(wcur, t, (c + 2) * KC);
  };
  for (int t = wid; t < T; t += nwv) {
    float off_h, off_w, mk;
    load_offsets(t, off_h, off_w, mk);     // first in the queue: loads come back in order
    load_w(wa, t, 0);                      // nothing they depend on: under the offsets' round trip
    if (1 < NCH) load_w(wb, t, KC);
    wave_lds_sync();                       // the table's readers of the previous tap are done
    coords(t, off_h, off_w, mk);
    wave_lds_sync();
    TAPS_STAMP();
    issue(v0, 0);
    if (1 < NCH) issue(v1, KC);
    if (2 < NCH) issue(v2, 2 * KC);
    if (3 < NCH) issue(v3, 3 * KC);
    TAPS_STAMP();
    for (int c = 0; c < NCH; c += DEPTH) {
      step(t, c, v0, A0, wa);
      TAPS_STAMP();
      if (c + 1 < NCH) step(t, c + 1, v1, A1, wb);
      if (c + 2 < NCH) step(t, c + 2, v2, A0, wa);
      if (c + 3 < NCH) step(t, c + 3, v3, A1, wb);
    }
  }
  wave_lds_sync();

  // ---- partial blocks meet in LDS: [pixel][output], one per wave ------------------------
  // accumulator map (16x16, block nb): column = 16 nb + i (output), row = 4 kg + reg (pixel)
  float *Rs = reinterpret_cast<float *>(s_t + wid * W_BYTES);
#pragma unroll
  for (int nb = 0; nb < 4; ++nb)
#pragma unroll
    for (int r = 0; r < 4; ++r) Rs[(4 * kg + r) * RROW + nb * 16 + i] = acc[nb][r];
  TAPS_STAMP();
  __syncthreads();
  TAPS_STAMP();
  TO *out = static_cast<TO *>(a.out);
  const TO *bias = static_cast<const TO *>(a.bias);
  const int nthr = nwv * 64;
  for (int e = tid; e < TAPS_PIX * 64; e += nthr) {
    // consecutive threads: consecutive outputs (channel-last) or consecutive pixels (NCHW)
    const int pix = a.out_nhwc ? e >> 6 : e & (TAPS_PIX - 1);
    const int c = a.out_nhwc ? e & 63 : e / TAPS_PIX;
    const long long p = tile * TAPS_PIX + pix;
    const int oc = n0 + c;
    if (p >= a.P || oc >= a.Cout) continue;
    // all partials requested at once (a loop over nwv would pay one LDS latency per wave),
    // added in wave order; absent waves add nothing (not even a zero: -0 stays -0)
    float part[TAPS_WAVES];
#pragma unroll
    for (int w = 0; w < TAPS_WAVES; ++w)
      part[w] = w < nwv ? *reinterpret_cast<const float *>(s_t + w * W_BYTES + pix * RROW * 4 + c * 4)
                        : 0.f;
    float sum = part[0];
#pragma unroll
    for (int w = 1; w < TAPS_WAVES; ++w) sum = w < nwv ? sum + part[w] : sum;
    if (bias) sum += to_f32(bias[oc]);
    if (a.relu) sum = fmaxf(sum, 0.f);
    if (a.out_nhwc) {
      out[p * a.Cout + oc] = from_f32<TO>(sum);
    } else {
      const int b = (int)(p / HWo);
      const int hw = (int)(p - (long long)b * HWo);
      out[((size_t)b * a.Cout + oc) * HWo + hw] = from_f32<TO>(sum);
    }
  }
  TAPS_STAMP();
#undef TAPS_STAMP
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

// W[Cout][Cin/groups][T] (TS) -> Wp[T][CoutP][Cp] (TD), zero padded.  Convolution groups
// (deform_conv_cuda.cpp:227-236: one addmm_ per group) become a block-diagonal dense
// weight: output o of group g = o / (Cout/groups) reads channels [g*Cin/groups,
// (g+1)*Cin/groups), every other entry of its row is zero, and the kernels stay as they are.
__device__ __forceinline__ long long grouped_weight_index(int o, int c, int t, int Cout, int Cin,
                                                          int T, int groups) {
  const int cig = Cin / groups, cog = Cout / groups;
  const int cg = c - (o / cog) * cig;
  if (cg < 0 || cg >= cig) return -1;
  return ((long long)o * cig + cg) * T + t;
}

// Element (t, o, c) of the fragment-major copy that dcn_taps_kernel reads: blocks of 16 outputs x
// 32 channels in the order the MFMA 16x16x32 takes them (lane = o % 16 + 16 * (c % 32 / 8), eight
// channels per lane), so that one load instruction of a wavefront reads 1 KB in a row.
__device__ __forceinline__ long long frag_major_index(int t, int o, int c, int CoutP, int Cp) {
  const long long block = ((long long)t * (Cp / 32) + c / 32) * (CoutP / 16) + o / 16;
  const int lane = (o & 15) + 16 * ((c & 31) >> 3);
  return (block * 64 + lane) * 8 + (c & 7);
}

// frag_off: where the fragment-major copy starts (elements of TD), 0 = none
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void pack_weight_kernel(
    const TS *__restrict__ w, TD *__restrict__ wp, int Cout, int Cin, int T,
    int CoutP, int Cp, int groups, long long frag_off) {
  const long long total = (long long)T * CoutP * Cp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int o = (int)((i / Cp) % CoutP);
    const int t = (int)(i / ((long long)Cp * CoutP));
    float v = 0.f;
    if (o < Cout && c < Cin) {
      const long long k = grouped_weight_index(o, c, t, Cout, Cin, T, groups);
      if (k >= 0) v = to_f32(w[k]);
    }
    wp[i] = from_f32<TD>(v);
    if (frag_off) wp[frag_off + frag_major_index(t, o, c, CoutP, Cp)] = from_f32<TD>(v);
  }
}

// W[Cout][Cin][T] float32 -> two bf16 planes Wp[T][CoutP][Cp]: hi = bf16(w), then
// lo = bf16(w - hi)  (cts_gemm_split_kernel, deform_conv_cts.hip)
__global__ __launch_bounds__(256) void pack_weight_split_kernel(
    const float *__restrict__ w, bf16_t *__restrict__ wp, int Cout, int Cin, int T,
    int CoutP, int Cp, int groups, long long frag_off) {
  const long long total = (long long)T * CoutP * Cp;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total;
       i += (long long)gridDim.x * 256) {
    const int c = (int)(i % Cp);
    const int o = (int)((i / Cp) % CoutP);
    const int t = (int)(i / ((long long)Cp * CoutP));
    float v = 0.f;
    if (o < Cout && c < Cin) {
      const long long k = grouped_weight_index(o, c, t, Cout, Cin, T, groups);
      if (k >= 0) v = w[k];
    }
    const bf16_t hi = from_f32<bf16_t>(v);
    const bf16_t lo = from_f32<bf16_t>(v - to_f32(hi));
    wp[i] = hi;
    wp[total + i] = lo;
    if (frag_off) {
      const long long k = frag_off + frag_major_index(t, o, c, CoutP, Cp);
      wp[k] = hi;
      wp[total + k] = lo;
    }
  }
}

// ---- host side ------------------------------------------------------------------

struct Plan {
  int Ho, Wo, T;
  int ta_dummy;
  int ta;        // arithmetic / LDS element type (ganet_dtype)
  int kc;        // channel chunk (elements)
  int Cp, CoutP;
  bool copy_x;   // a channel-last (and/or converted, padded) copy of x is made
  bool cts;      // contract-then-sample path (deform_conv_cts.hip)
  bool split;    // ... with float32 tensors contracted as three bf16 passes
  bool fsplit;   // the fused kernel with float32 tensors contracted as three bf16 passes
  bool stream;   // ... as one persistent producer/consumer launch (deform_conv_stream.hip)
  bool ring;     // ... as one launch with Y kept in L2, tap-major (deform_conv_ring.hip)
  bool taps;     // small batches: the taps of a pixel tile on different wavefronts (dcn_taps_kernel)
  bool keep_y;   // the whole batch's per-tap products stay in the workspace (GANET_DCN_ALGO_KEEP_Y)
  int nb;        // images per chunk of the contract-then-sample pair
  size_t xh_bytes, wp_bytes, y_bytes;
  size_t wp_frag_off;   // the fragment-major copy of the packed weights (bytes into them), 0 = none
};

size_t dtype_size(int dt) { return dt == GANET_F32 ? 4 : 2; }

// tiles (16 pixels x 64 outputs) up to which algorithm 5 is chosen by itself; GANET_TAPS_MAX_TILES
// is an A/B knob, read once per process (0 = never)
long long taps_max_tiles() {
  static const long long v = [] {
    const char *e = getenv("GANET_TAPS_MAX_TILES");
    return e ? atoll(e) : 256ll;
  }();
  return v;
}

int make_plan(const ganet_dcn_params *p_in, Plan *pl) {
  if (!p_in) return GANET_ERR_NULL;
  // `algo` carries the algorithm in its low byte and GANET_DCN_ALGO_KEEP_Y above it
  ganet_dcn_params q = *p_in;
  const bool keep_y = (q.algo & GANET_DCN_ALGO_KEEP_Y) != 0;
  if (q.algo < 0 || (q.algo & ~(0xff | GANET_DCN_ALGO_KEEP_Y))) return GANET_ERR_BAD_SHAPE;
  q.algo &= 0xff;
  const ganet_dcn_params *p = &q;
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
  if (!ganet_dcn_ranges_ok(p)) return GANET_ERR_UNSUPPORTED;
  pl->Ho = ganet_dcn_out_size(p->H, p->pH, p->dH, p->kH, p->sH);
  pl->Wo = ganet_dcn_out_size(p->W, p->pW, p->dW, p->kW, p->sW);
  if (pl->Ho < 1 || pl->Wo < 1) return GANET_ERR_TOO_SMALL;
  if (p->offset_is_points &&
      (p->kH != 1 || p->dW != 1 || p->pW != (p->kW - 1) / 2 || !(p->kW & 1)))
    return GANET_ERR_BAD_SHAPE;
  if ((long long)p->B * pl->Ho >= (1ll << 31) ||
      (long long)p->B * pl->Ho * pl->Wo >= (1ll << 31))
    return GANET_ERR_UNSUPPORTED;
  pl->T = p->kH * p->kW;
  pl->ta_dummy = 0;
  if (p->mfma_bf16 < 0 || p->mfma_bf16 > 2) return GANET_ERR_BAD_SHAPE;
  pl->ta = (p->dtype == GANET_F32 && p->mfma_bf16 == 1) ? GANET_BF16 : p->dtype;
  // weight chunk (channels): 512 B per output row when the channel count allows,
  // else 256 B / 128 B; channels are zero-padded to the smallest chunk otherwise
  const int kc_big = pl->ta == GANET_F32 ? 128 : 256;
  const int kc_small = kc_big / 4;
  const int cpg = p->Cin / p->deformable_groups;
  if (cpg % kc_big == 0) {
    pl->kc = kc_big;
    pl->Cp = p->Cin;
  } else if (cpg % (kc_big / 2) == 0) {
    pl->kc = kc_big / 2;
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
  {
    // a tile's 32-bit buffer window spans every image the tile touches and must
    // stay below 2 GiB (see OOB_OFFSET in the kernel)
    const unsigned long long img_bytes =
        (unsigned long long)p->H * p->W * pl->Cp * dtype_size(pl->ta);
    const unsigned long long span = (unsigned long long)TM_MAX / ((unsigned long long)pl->Ho * pl->Wo) + 2;
    if (img_bytes * span >= 0x7fffffffull) return GANET_ERR_UNSUPPORTED;
  }
  pl->copy_x = p->x_layout != GANET_NHWC || pl->Cp != p->Cin || pl->ta != p->dtype;
  pl->xh_bytes = pl->copy_x ? ganet_align_up((size_t)p->B * p->H * p->W * pl->Cp *
                                                 dtype_size(pl->ta), 256)
                            : 0;
  pl->wp_bytes = ganet_align_up(
      (size_t)pl->T * pl->CoutP * pl->Cp * dtype_size(pl->ta), 256);
  pl->wp_frag_off = 0;
  // Sampling and contraction commute; contract first when that shrinks the rows
  // that have to be gathered and blended (Cout < Cin, as in the LFA).
  pl->cts = cts_applicable(p, pl->Cp, pl->CoutP) && p->algo != 1 && p->algo != 5;
  if (p->algo < 0 || p->algo > 5) return GANET_ERR_BAD_SHAPE;
  if (p->algo == 2 && !pl->cts) return GANET_ERR_UNSUPPORTED;
  pl->split = pl->cts && pl->ta == GANET_F32 && cts_split(p, pl->Cp);
  pl->fsplit = !pl->cts && pl->ta == GANET_F32 && p->mfma_bf16 == 0;
  // Small batches (algorithm 5): one launch whose wavefronts split the taps.  Chosen by itself
  // while the call is a single round of workgroups (at most taps_max_tiles() tiles of 16 pixels x
  // 64 outputs) and only where it reads the packed weights the other choice would read (plain
  // planes for 16-bit arithmetic, hi/lo planes for float32), so that weights packed for a
  // shape at one batch size serve it at any other.
  {
    const long long tiles = ((long long)p->B * pl->Ho * pl->Wo + TAPS_PIX - 1) / TAPS_PIX * (pl->CoutP / TN);
    const bool can = p->deformable_groups == 1 && !keep_y && !p->debug &&
                     (pl->ta != GANET_F32 || p->mfma_bf16 == 0);
    const bool same_pack = pl->ta != GANET_F32 || pl->split || pl->fsplit;
    if (p->algo == 5 && !can) return GANET_ERR_UNSUPPORTED;
    pl->taps = can && (p->algo == 5 || (p->algo == 0 && same_pack && tiles <= taps_max_tiles()));
    if (pl->taps) {
      pl->cts = pl->split = false;
      pl->fsplit = pl->ta == GANET_F32;
    }
  }
  pl->stream = pl->cts && p->algo == 3 &&
               stream_applicable(p, pl->ta, pl->Cp, pl->CoutP, pl->Ho, pl->Wo);
  if (p->algo == 3 && !pl->stream) return GANET_ERR_UNSUPPORTED;
  pl->ring = pl->cts && p->algo == 4 &&
             ring_applicable(p, pl->ta, pl->Cp, pl->CoutP, pl->Ho, pl->Wo);
  if (p->algo == 4 && !pl->ring) return GANET_ERR_UNSUPPORTED;
  // keeping Y for the backward: float32 tensors in the default arithmetic (the backward forms
  // the same three-pass products), one chunk for the whole batch
  pl->keep_y = keep_y;
  if (keep_y && (!pl->split || pl->stream || pl->ring || p->Cout % 64)) return GANET_ERR_UNSUPPORTED;
  pl->nb = 0;
  if (pl->cts && !pl->stream && !pl->ring) {
    pl->nb = keep_y ? p->B : cts_chunk_images(p, pl->CoutP, pl->ta, pl->split);
    const unsigned long long img_y = (unsigned long long)p->H * p->W * pl->T * pl->CoutP *
                                     dtype_size(pl->ta);
    if (img_y * (unsigned long long)pl->nb >= 0x7fffffffull) return GANET_ERR_UNSUPPORTED;
  }
  // 16-bit planes (plain, or the hi/lo pair of float32 tensors) are packed twice: [tap][Cout][Cin]
  // for algorithms 1-4 and fragment-major for algorithm 5 - whatever the batch size of `p`, so
  // that weights packed for a shape at one batch size serve it at any other
  if (pl->ta != GANET_F32 || pl->split || pl->fsplit) {
    pl->wp_frag_off = pl->wp_bytes;
    pl->wp_bytes *= 2;
  }
  pl->y_bytes = pl->stream ? stream_workspace_bytes(p, pl->CoutP)
                : pl->ring ? ring_workspace_bytes(p)
                           : (pl->cts ? cts_y_bytes(p, pl->CoutP, pl->ta, pl->nb) : 0);
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
                     p->Cin, pl.T, pl.CoutP, pl.Cp, p->groups,
                     (long long)(pl.wp_frag_off / sizeof(TD)));
}

template <typename TA, typename TO, int KCB, bool SPLIT = false>
void launch_nw(const Plan &pl, const DcnArgs &a, hipStream_t s) {
  // 8 waves (256 pixels) per workgroup when that still gives every CU work;
  // smaller workgroups for small batches so the grid covers the chip.
  const long long waves = (a.P + 31) / 32;
  const unsigned ny = (unsigned)(pl.CoutP / TN);
  if (waves >= 8 * 256) {
    dim3 grid((unsigned)((a.P + 255) / 256), ny);
    hipLaunchKernelGGL((dcn_fused_kernel<TA, TO, KCB, 8, SPLIT>), grid, dim3(512), 0, s, a);
  } else if (waves >= 4 * 128) {
    dim3 grid((unsigned)((a.P + 127) / 128), ny);
    hipLaunchKernelGGL((dcn_fused_kernel<TA, TO, KCB, 4, SPLIT>), grid, dim3(256), 0, s, a);
  } else {
    dim3 grid((unsigned)((a.P + 63) / 64), ny);
    hipLaunchKernelGGL((dcn_fused_kernel<TA, TO, KCB, 2, SPLIT>), grid, dim3(128), 0, s, a);
  }
}

template <typename TA, typename TO, bool SPLIT = false>
void launch_main(const Plan &pl, const DcnArgs &a, hipStream_t s) {
  constexpr int KC_BIG = sizeof(TA) == 4 ? 128 : 256;
  if (pl.kc == KC_BIG)
    launch_nw<TA, TO, KC_BIG, SPLIT>(pl, a, s);
  else if (pl.kc == KC_BIG / 2)
    launch_nw<TA, TO, KC_BIG / 2, SPLIT>(pl, a, s);
  else
    launch_nw<TA, TO, KC_BIG / 4, SPLIT>(pl, a, s);
}

}  // namespace

extern "C" int ganet_dcn_out_size(int in, int pad, int dilation, int k,
                                  int stride) {
  if (stride <= 0) return 0;
  // 64-bit: any int32 arguments are a defined computation (the reference computes in Python)
  const long long num = (long long)in + 2ll * pad - ((long long)dilation * ((long long)k - 1) + 1);
  if (num < 0) return 0;  // Python floor division would give <= 0 here
  const long long out = num / stride + 1;
  return out > 0x7fffffffll ? 0 : (int)out;
}

extern "C" size_t ganet_deform_conv_workspace_bytes(const ganet_dcn_params *p) {
  Plan pl;
  if (make_plan(p, &pl) != GANET_OK) return 0;
  return pl.xh_bytes + pl.y_bytes + (p->weight_is_packed ? 0 : pl.wp_bytes);
}

extern "C" int ganet_deform_conv_algorithm(const ganet_dcn_params *p) {
  Plan pl;
  if (!p || make_plan(p, &pl) != GANET_OK) return 0;
  return pl.taps ? 5 : (pl.ring ? 4 : (pl.stream ? 3 : (pl.cts ? 2 : 1)));
}

extern "C" size_t ganet_deform_conv_status_offset(const ganet_dcn_params *p) {
  Plan pl;
  if (!p || make_plan(p, &pl) != GANET_OK || !(pl.stream || pl.ring)) return (size_t)-1;
  return pl.xh_bytes;   // the control block's first word (the status) leads algorithm 3's and 4's scratch
}

extern "C" size_t ganet_deform_conv_y_offset(const ganet_dcn_params *p) {
  Plan pl;
  if (!p || make_plan(p, &pl) != GANET_OK || !pl.keep_y) return (size_t)-1;
  return pl.xh_bytes;
}

extern "C" size_t ganet_deform_conv_packed_weight_bytes(const ganet_dcn_params *p) {
  Plan pl;
  if (make_plan(p, &pl) != GANET_OK) return 0;
  return pl.wp_bytes;
}

namespace {
int pack_weight_dispatch(const ganet_dcn_params *p, const Plan &pl,
                         const void *weight, void *wp, hipStream_t s) {
  if (pl.split || pl.fsplit) {
    const long long total = (long long)pl.T * pl.CoutP * pl.Cp;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_weight_split_kernel, dim3(blocks), dim3(256), 0, s,
                       static_cast<const float *>(weight), static_cast<bf16_t *>(wp),
                       p->Cout, p->Cin, pl.T, pl.CoutP, pl.Cp, p->groups,
                       (long long)(pl.wp_frag_off / 2));
  } else if (p->dtype == GANET_F32 && pl.ta == GANET_F32)
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
  const size_t need = pl.xh_bytes + pl.y_bytes + (p->weight_is_packed ? 0 : pl.wp_bytes);
  if (need > 0 && (workspace_bytes < need || !workspace))
    return GANET_ERR_WORKSPACE;

  hipStream_t s = static_cast<hipStream_t>(stream);
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  void *xh = pl.copy_x ? static_cast<void *>(ws) : nullptr;
  void *ybuf = ws + pl.xh_bytes;
  const void *wp = p->weight_is_packed
                       ? weight
                       : static_cast<const void *>(ws + pl.xh_bytes + pl.y_bytes);

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
      pack_weight_dispatch(p, pl, weight, ws + pl.xh_bytes + pl.y_bytes, s) != GANET_OK)
    return GANET_ERR_HIP;

  if (pl.ring)
    return ring_forward(pl.copy_x ? xh : x, offset, mask, wp, bias, out, p, pl.ta, pl.Cp, ybuf, s);
  if (pl.stream)
    return stream_forward(pl.copy_x ? xh : x, offset, mask, wp, bias, out, p, pl.ta, pl.Cp,
                          pl.CoutP, ybuf, s);
  if (pl.cts)
    return cts_forward(pl.copy_x ? xh : x, offset, mask, wp, bias, out, p, pl.ta, pl.kc,
                       pl.Cp, pl.CoutP, pl.Ho, pl.Wo, ybuf, pl.nb, s);

  DcnArgs a;
  a.xh = pl.copy_x ? xh : x;
  a.offset = offset;
  a.mask = mask;
  a.wp = wp;
  a.wp_frag = static_cast<const unsigned char *>(wp) + pl.wp_frag_off;
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
  a.ablate = p->debug;
  a.P = (long long)p->B * pl.Ho * pl.Wo;

  if (pl.taps) {
    const int nwv = pl.T < TAPS_WAVES ? pl.T : TAPS_WAVES;
    const dim3 grid((unsigned)((a.P + TAPS_PIX - 1) / TAPS_PIX), (unsigned)(pl.CoutP / TN)), block(nwv * 64);
    if (pl.ta == GANET_F32)
      hipLaunchKernelGGL((dcn_taps_kernel<float, float, true>), grid, block, 0, s, a);
    else if (pl.ta == GANET_BF16 && p->dtype == GANET_F32)
      hipLaunchKernelGGL((dcn_taps_kernel<bf16_t, float, false>), grid, block, 0, s, a);
    else if (pl.ta == GANET_BF16)
      hipLaunchKernelGGL((dcn_taps_kernel<bf16_t, bf16_t, false>), grid, block, 0, s, a);
    else
      hipLaunchKernelGGL((dcn_taps_kernel<f16_t, f16_t, false>), grid, block, 0, s, a);
    return ganet_check_launch();
  }
  if (pl.ta == GANET_F32 && pl.fsplit)
    launch_main<float, float, true>(pl, a, s);
  else if (pl.ta == GANET_F32)
    launch_main<float, float>(pl, a, s);
  else if (pl.ta == GANET_BF16 && p->dtype == GANET_F32)
    launch_main<bf16_t, float>(pl, a, s);
  else if (pl.ta == GANET_BF16)
    launch_main<bf16_t, bf16_t>(pl, a, s);
  else
    launch_main<f16_t, f16_t>(pl, a, s);
  return ganet_check_launch();
}

#ifdef GANET_TAPS_STAMPS
extern "C" int ganet_debug_taps_stamps(long long *host48) {
  return hipMemcpyFromSymbol(host48, HIP_SYMBOL(g_taps_stamps), 48 * sizeof(long long)) == hipSuccess
             ? GANET_OK
             : GANET_ERR_HIP;
}
#endif
