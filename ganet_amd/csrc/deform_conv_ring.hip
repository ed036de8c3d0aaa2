// Deformable convolution forward, contract-then-sample form with the per-tap products kept
// on chip ("algorithm 4") for gfx950 (MI355X).
//
// Algorithm 2 (csrc/deform_conv_cts.hip) replaces deformable_im2col + addmm_
// (mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:222-236) by  Y_t = X * W_t^T  and
// out = sum_t sample(Y_t), and pays for it with Y's round trip through the fabric: 918 of the
// 1272 MB one LFA call moves at 128 images.  Here Y never leaves the XCD that formed it:
//
//   * a TEAM of 32 workgroups (512 threads each, two workgroups per CU, so two teams per
//     XCD) owns one image at a time; the 256 waves of a team own 16 pixels each and keep
//     those pixels' x (16 x Cin) in registers as MFMA operands for the image's whole life;
//   * Y is formed TAP-MAJOR: in step n every wave multiplies its x block with W_t (32 KB,
//     staged in LDS as ready-made 16x16x32 operand images) and stores its 16 pixels of plane
//     t (128 B per pixel) into slot n % 2 of the team's ring: two planes of H*W*128 B, the
//     same 1 MB of addresses for every tap and every image, so the lines are dirtied again
//     in L2 instead of being written back;
//   * in the same step the wave gathers plane n-1 from the other slot (8 lanes x 16 B per
//     pixel, a corner outside the map reads 0 through the buffer bounds check) and adds it
//     into the fp32 sums of the 16 output pixels it owns; sums, x and coordinates stay in
//     registers / LDS across the T taps.
//
// Schedule: static.  A workgroup reads its XCD from the hardware register and draws a ticket
// from that XCD's counter: ticket / 32 is its team, ticket % 32 its rank; team g takes
// images g, g + 16, ...  There are no queues and no claims.  One barrier per step and team
// (a monotonic counter in the team's own cache line: one add per workgroup, polled by one
// lane per workgroup) orders "plane n is complete" before its gathers and "plane n-2 has
// been read" before slot n % 2 is overwritten; the MFMAs of step n+1 run between a
// workgroup's arrival and its wait.
// Visibility: producer and consumer share one L2 by construction (teams are formed per
// XCC id), Y is stored by plain stores (they stay in L2), every storing wave drains them
// (s_waitcnt vmcnt(0)) before its workgroup arrives, and every load of Y is an sc1 load
// (served by L2, never by the CU's L1, which may hold the slot's previous plane).
// Every wait is bounded; a wait that runs out leaves a non-zero status word and every wave
// leaves.  The last workgroup to finish checks that every image was taken by some team.
//
// Arithmetic of the sampling position and weights: deform_conv_cuda_kernel.cu:84-115,
// 218-229 (and :604-627 for the DCNv2 mask), unchanged; blend order as in algorithm 2.

#include "dcn_device.h"
#include <cstdlib>

namespace ganet_dcn {

namespace {

constexpr int RW = 8;                        // waves per workgroup
constexpr int RNT = RW * 64;
constexpr int TEAM_WGS = 32;                 // workgroups per team
constexpr int TEAMS_PER_XCD = 2;
constexpr int RING_XCDS = 8;
constexpr int NTEAMS = RING_XCDS * TEAMS_PER_XCD;
constexpr int WPX = 16;                      // pixels per wave
constexpr int TEAM_PX = TEAM_WGS * RW * WPX; // 4096 pixels per plane
constexpr int NSLOT = 2;
constexpr int PXB = 128;                     // bytes per pixel of a plane (64 outputs)
constexpr int RING_TMAX = 16;                // taps
constexpr unsigned long long RING_TIMEOUT_TICKS = 200000000ull;  // 2 s of the 100 MHz clock
constexpr int AUX_SC1 = 16;

// control block (32-bit words); hot words sit in 128-byte lines of their own
enum {
  RL = 32,
  RCTL_ERROR = 0,
  RCTL_FINISHED = 1 * RL,
  RCTL_IMAGES = 2 * RL,
  RCTL_JOIN = 3 * RL,                        // [8] one line per XCC id
  RCTL_BAR = RCTL_JOIN + RING_XCDS * RL,     // [NTEAMS] one line per team
  RCTL_STAMPS = RCTL_BAR + NTEAMS * RL,      // debug builds: 8 x 64-bit phase sums
  RCTL_WORDS = RCTL_STAMPS + RL
};
constexpr size_t RING_CTL_BYTES = 4096;
static_assert(RCTL_WORDS * 4 <= RING_CTL_BYTES, "control block");

struct RingArgs {
  const void *x;        // [B][H*W][KH] channel-last, 16-bit arithmetic dtype
  const void *wq;       // [T][64][KH]
  unsigned char *ring;  // [NTEAMS][NSLOT][plane_bytes]
  const float *offset;  // [B][2T][H][W]
  const float *mask;    // [B][T][H][W] or nullptr
  const void *bias;     // [Cout] or nullptr
  void *out;
  unsigned int *ctl;
  int B, H, W, Cout, T;
  int kW, pH, pW, dH, dW;
  int out_nhwc, relu, points;
  unsigned plane_bytes;
  int grid;             // workgroups launched (RING_XCDS * TEAMS_PER_XCD * TEAM_WGS)
  int debug;            // timing experiments: bit0 nothing produced, bit1 nothing sampled,
                        // bit2 no waiting at the team barrier (results then undefined)
};

typedef __attribute__((address_space(1))) unsigned int rgu32;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;

__device__ __forceinline__ unsigned int rflag_load(const unsigned int *p) {
  return __hip_atomic_load((const rgu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int rflag_add(unsigned int *p, unsigned int v) {
  return __hip_atomic_fetch_add((rgu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void rflag_store(unsigned int *p, unsigned int v) {
  __hip_atomic_store((rgu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ void rwave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// workgroup barrier that waits for LDS traffic only: vector loads stay in flight across it
__device__ __forceinline__ void wg_barrier_lds() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// ... and one that first drains every vector memory operation of the wave (stores of Y,
// gathers, LDS-DMA of the next weights)
__device__ __forceinline__ void wg_barrier_drain() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename TA>
struct Mma16;
template <>
struct Mma16<bf16_t> {
  typedef bf16x8 frag;
  __device__ static __forceinline__ f32x4 mma(const frag &a, const frag &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <>
struct Mma16<f16_t> {
  typedef f16x8 frag;
  __device__ static __forceinline__ f32x4 mma(const frag &a, const frag &b, const f32x4 &c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
};

// one 16-byte piece of a corner row, weight w, into fp32 sums (as algorithm 2)
template <typename TY>
__device__ __forceinline__ void racc_piece(const u32x4 &v, float w, f32x2 *acc);
template <>
__device__ __forceinline__ void racc_piece<bf16_t>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 d = {__uint_as_float(v[j] << 16), __uint_as_float(v[j] & 0xffff0000u)};
    acc[j] = d * w + acc[j];
  }
}
template <>
__device__ __forceinline__ void racc_piece<f16_t>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int u = v[j];
    const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
    const _Float16 hi = __builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    const f32x2 d = {(float)lo, (float)hi};
    acc[j] = d * w + acc[j];
  }
}

// Sampling position of output pixel (ho, wo), tap t, as one table entry:
//   x = byte offset of corner (h_low, w_low) inside a plane (a multiple of 128, negative when
//       h_low or w_low is -1) | validity of the four corners in bits 0..3,  y = lh,  z = lw,
//   w = mask value (0 for a sample outside the map: every corner then reads 0 anyway).
template <typename A>
__device__ __forceinline__ uint4 ring_entry(const A &a, int ho, int wo, int t, float off_h,
                                            float off_w, float mk) {
  const int ti = t / a.kW, tj = t - ti * a.kW;
  if (a.points)  // dcn_offset = pts - base  (lanepoints_conv.py:63-74,170)
    off_w = off_w - (float)(tj - a.pW);
  const float h_im = (float)(ho - a.pH + ti * a.dH) + off_h;
  const float w_im = (float)(wo - a.pW + tj * a.dW) + off_w;
  unsigned e0 = 0;
  float lh = 0.f, lw = 0.f, m = 0.f;
  if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
    const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
    const int h_high = h_low + 1, w_high = w_low + 1;
    lh = h_im - (float)h_low;
    lw = w_im - (float)w_low;
    m = mk;
    const bool hl = h_low >= 0, hh_ok = h_high <= a.H - 1;
    const bool wl = w_low >= 0, wh_ok = w_high <= a.W - 1;
    const unsigned vb = (hl && wl ? 1u : 0u) | (hl && wh_ok ? 2u : 0u) | (hh_ok && wl ? 4u : 0u) |
                        (hh_ok && wh_ok ? 8u : 0u);
    e0 = (unsigned)((h_low * a.W + w_low) * PXB) | vb;
  }
  return make_uint4(e0, __float_as_uint(lh), __float_as_uint(lw), __float_as_uint(m));
}

#ifdef GANET_RING_DEBUG
#define RDEBUG(a, bit) ((a).debug & (bit))
#define RSTAMP_DECL unsigned long long rs_t = __builtin_amdgcn_s_memtime(), rs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define RSTAMP(ph)                                                  \
  do {                                                              \
    const unsigned long long rs_n = __builtin_amdgcn_s_memtime();   \
    rs_acc[ph] += rs_n - rs_t;                                      \
    rs_t = rs_n;                                                    \
  } while (0)
#else
#define RDEBUG(a, bit) 0
#define RSTAMP_DECL
#define RSTAMP(ph) do {} while (0)
#endif

// the wave's s_waitcnt forms the compiler's own wait bookkeeping understands (gfx9 encoding:
// vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14)
#define RING_WAIT_ALL() __builtin_amdgcn_s_waitcnt(0x0070)        /* vmcnt(0) lgkmcnt(0) */
#define RING_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f)        /* lgkmcnt(0) */
#define RING_BARRIER() asm volatile("s_barrier" ::: "memory")

__device__ __forceinline__ int rsgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
// a wave-uniform value the optimiser may not look through: what is computed from it stays where
// it is written (nothing is hoisted out of the step loop into registers that then spill)
__device__ __forceinline__ int ropaque(int v) {
  asm volatile("" : "+s"(v));
  return v;
}
template <typename P>
__device__ __forceinline__ P *ropaque_ptr(P *p) {
  asm volatile("" : "+s"(p));
  return p;
}
template <typename P>
__device__ __forceinline__ P *rsgpr_ptr(P *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (P *)(((unsigned long long)hi << 32) | lo);
}

template <typename TA, typename TO, int KH>
__global__ __launch_bounds__(RNT, 4) void lfa_ring_kernel(RingArgs a) {
  constexpr int NKS = KH / 32;               // 32-deep MFMA steps
  constexpr int NIMG = 4 * NKS;              // 1-KB operand images of one tap's weights
  constexpr int W_BYTES = NIMG * 1024;
  constexpr int TROW = RING_TMAX + 1;        // entries per pixel row (+1: rows start on different banks)
  constexpr int TAB_BYTES = RW * WPX * TROW * 16;
  typedef typename Mma16<TA>::frag frag_t;
  // one array: weights, the waves' coordinate tables, a few words of workgroup state
  __shared__ __attribute__((aligned(1024))) unsigned char smem[W_BYTES + TAB_BYTES + 64];
  int *s_info = reinterpret_cast<int *>(smem + W_BYTES + TAB_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = rsgpr(tid >> 6);
  unsigned int *ctl = a.ctl;

  // ---- join: XCD, team, rank ------------------------------------------------------------
  if (tid == 0) {
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    xid &= 7u;
    const unsigned ticket = rflag_add(&ctl[RCTL_JOIN + xid * RL], 1u);
    s_info[0] = (int)xid;
    s_info[1] = (int)ticket;
    s_info[2] = 0;                           // failure word of the workgroup
  }
  __syncthreads();
  const int xcc = rsgpr(s_info[0]);
  const int ticket = rsgpr(s_info[1]);
  const int team_x = ticket / TEAM_WGS, rank = ticket - team_x * TEAM_WGS;
  const int team = xcc * TEAMS_PER_XCD + team_x;
  const bool seated = team_x < TEAMS_PER_XCD;  // else: more workgroups on this XCD than the schedule has
  const int HW = a.H * a.W;
  const int T = a.T;
  const int nimg = (!seated || team >= a.B) ? 0 : (a.B - team + NTEAMS - 1) / NTEAMS;
  const int NP = rsgpr(nimg * T);
  bool failed = false;
  if (!seated && tid == 0) rflag_store(&ctl[RCTL_ERROR], 2u);

  if (NP > 0) {
    unsigned int *bar = rsgpr_ptr(ctl + RCTL_BAR + team * RL);
    // pixels: the wave's 16 as MFMA columns (lane & 15) and as two gather groups of 8
    const int px0 = rsgpr((rank * RW + wid) * WPX);
    const bool active = px0 < HW;            // wave-uniform
    const int mc = lane & 15, mg = lane >> 4;  // MFMA: column (pixel), k group / row group
    const int pxg = lane >> 3, pj = lane & 7;  // gather: pixel of the group, 16-byte piece
    unsigned char *ring = rsgpr_ptr(a.ring + (size_t)team * NSLOT * a.plane_bytes);
    const __amdgpu_buffer_rsrc_t ysrc =
        __builtin_amdgcn_make_buffer_rsrc(ring, 0, (int)(NSLOT * a.plane_bytes), 0x00020000);
    const int row_bytes = a.W * PXB;
    const unsigned plane_bytes = a.plane_bytes;
    const uint4 *tab_rd = reinterpret_cast<const uint4 *>(smem + W_BYTES) + wid * (WPX * TROW) + pxg * TROW;
    uint4 *tab_wr = reinterpret_cast<uint4 *>(smem + W_BYTES) + wid * (WPX * TROW) + pxg * TROW;
    const unsigned char *xbytes = static_cast<const unsigned char *>(a.x);
    const unsigned char *wq = static_cast<const unsigned char *>(a.wq);
    const size_t img_x_bytes = (size_t)HW * KH * 2;

    // per-lane byte offsets (32 bit) against wave-uniform bases
    const int pm = px0 + mc < HW ? px0 + mc : HW - 1;
    const unsigned x_off = (unsigned)pm * (KH * 2) + (unsigned)mg * 16u;
    const unsigned y_off = (unsigned)(px0 + mc) * PXB + (unsigned)mg * 16u;
    const bool y_ok = px0 + mc < HW;
    // weights of tap t into LDS as operand images: image (m, ks) holds, for lane l, the 16
    // bytes W_t[row(m, l & 15)][32 ks + 8 (l >> 4) ..+8].  Row r = 4 g + i of tile m is output
    // 32 (m >> 1) + 8 g + 4 (m & 1) + i: accumulator register i of lane (pixel, g) in tiles
    // 2h, 2h + 1 then holds outputs 32 h + 8 g .. +8 of that pixel, 16 contiguous bytes of Y.
    const unsigned w_off = (unsigned)((8 * ((lane & 15) >> 2) + (lane & 3)) * KH + (lane >> 4) * 8) * 2u;
    auto stage_weights = [&](int t) {
      t = ropaque(t);
#pragma unroll
      for (int j = 0; j < NIMG / RW; ++j) {
        const int ii = j * RW + wid;
        const int m = ii / NKS, ks = ii - m * NKS;
        const unsigned char *base =
            ropaque_ptr(wq + ((size_t)(t * 64 + 32 * (m >> 1) + 4 * (m & 1)) * KH + ks * 32) * 2);
        __builtin_amdgcn_global_load_lds((const glb_void *)(base + w_off), (lds_void *)(smem + ii * 1024),
                                         16, 0, 0);
      }
    };

    u32x4 xb[NKS];
    auto load_x = [&](int img) {
      img = ropaque(img);
      const unsigned char *base = ropaque_ptr(xbytes + (size_t)img * img_x_bytes);
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) xb[ks] = ldg128(base + x_off + ks * 64);
    };

    // table of image img: lane (pxg, pj) forms taps pj, pj + 8 of pixels pxg and 8 + pxg
    // the image-edge paths read their arguments again from the kernel argument segment (scalar
    // loads, once per image) instead of holding them in SGPRs through the step loop
    typedef const __attribute__((address_space(4))) RingArgs *cargs_t;
    auto form_table = [&](int img) {
      img = ropaque(img);
      cargs_t ap = ropaque_ptr((cargs_t)__builtin_amdgcn_kernarg_segment_ptr());
      struct {
        const float *offset, *mask;
        int H, W, T, kW, pH, pW, dH, dW, points;
      } c = {ap->offset, ap->mask, ap->H, ap->W, ap->T, ap->kW, ap->pH, ap->pW, ap->dH, ap->dW, ap->points};
      const int T = c.T, HW = c.H * c.W;
      const float *offs = c.offset + (size_t)img * 2 * T * HW;
      const float *msk = c.mask ? c.mask + (size_t)img * T * HW : nullptr;
      const int pxo = ropaque(px0);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int p = pxo + st * 8 + pxg;
        const bool valid = p < HW;
        const int ho = valid ? p / c.W : 0;
        const int wo = valid ? p - ho * c.W : 0;
        for (int t = pj; t < T; t += 8) {
          uint4 e = make_uint4(0u, 0u, 0u, 0u);
          if (valid) {
            const unsigned o = (unsigned)(2 * t * HW + p);
            const float off_h = ldg_f32(offs + o);
            const float off_w = ldg_f32(offs + o + HW);
            const float mk = msk ? ldg_f32(msk + (unsigned)(t * HW + p)) : 1.f;
            e = ring_entry(c, ho, wo, t, off_h, off_w, mk);
          }
          tab_wr[st * 8 * TROW + t] = e;
        }
      }
      rwave_sync_lds();
    };

    f32x2 oacc[2][4];
    auto zero_out = [&]() {
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int j = 0; j < 4; ++j) oacc[st][j] = f32x2{0.f, 0.f};
    };
    auto store_out = [&](int img) {
      img = ropaque(img);
      cargs_t ap = ropaque_ptr((cargs_t)__builtin_amdgcn_kernarg_segment_ptr());
      struct {
        const void *bias;
        void *out;
        int H, W, Cout, out_nhwc, relu;
      } c = {ap->bias, ap->out, ap->H, ap->W, ap->Cout, ap->out_nhwc, ap->relu};
      const int HW = c.H * c.W;
      const TO *bias = static_cast<const TO *>(c.bias);
      TO *out = static_cast<TO *>(c.out);
      const int c0 = pj * 8;
      const bool fast = c.out_nhwc && c.Cout == 64 && sizeof(TO) == 2;
      const int pxo = ropaque(px0);
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const int p = pxo + st * 8 + pxg;
        if (p >= HW) continue;
        float r[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          r[2 * j] = oacc[st][j].x;
          r[2 * j + 1] = oacc[st][j].y;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          if (bias && c0 + e < c.Cout) r[e] += to_f32(bias[c0 + e]);
          if (c.relu) r[e] = fmaxf(r[e], 0.f);
        }
        const size_t pg = (size_t)img * HW + p;
        if (fast) {
          u32x4 pk;
          TO tmp[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) tmp[e] = from_f32<TO>(r[e]);
          __builtin_memcpy(&pk, tmp, 16);
          *(GANET_GLOBAL u32x4 *)(out + pg * 64 + c0) = pk;
        } else if (c.out_nhwc) {
          for (int e = 0; e < 8; ++e)
            if (c0 + e < c.Cout) out[pg * c.Cout + c0 + e] = from_f32<TO>(r[e]);
        } else {
          for (int e = 0; e < 8; ++e)
            if (c0 + e < c.Cout) out[((size_t)img * c.Cout + c0 + e) * HW + p] = from_f32<TO>(r[e]);
        }
      }
    };

    RSTAMP_DECL;
    // ---- prologue: the first image's weights, x and coordinates -----------------------
    stage_weights(0);
    if (active) {
      load_x(team);
      form_table(team);
    }
    zero_out();
    RING_WAIT_ALL();
    RING_BARRIER();
    RSTAMP(0);

    int t = 0, img = team;                   // tap and image of plane n
    for (int n = 0; n <= NP; ++n) {
      const bool do_mfma = n < NP;
      const bool do_gather = n >= 1;
      const int tp = t == 0 ? T - 1 : t - 1; // tap of plane n - 1
      const bool img_edge = do_gather && t == 0;   // plane n - 1 closed its image

      // the poll of barrier n-1 goes out ahead of the MFMAs
      unsigned seen = 0;
      if (do_gather && wid == 0 && lane == 0) seen = rflag_load(bar);

      // ---- plane n = x_block * W_t^T on the matrix cores --------------------------------
      u32x4 ypk[2];
      if (do_mfma && active && !RDEBUG(a, 1)) {
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        const u32x4 *wimg = reinterpret_cast<const u32x4 *>(smem) + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) {
          const frag_t fb = __builtin_bit_cast(frag_t, xb[ks]);
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            const frag_t fa = __builtin_bit_cast(frag_t, wimg[(m * NKS + ks) * 64]);
            acc[m] = Mma16<TA>::mma(fa, fb, acc[m]);
          }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          TA v[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = from_f32<TA>(acc[2 * h][i]);
            v[4 + i] = from_f32<TA>(acc[2 * h + 1][i]);
          }
          __builtin_memcpy(&ypk[h], v, 16);
        }
        // the image's last tap is done: the next image's x takes the registers
        if (t == T - 1 && n + 1 < NP) load_x(img + NTEAMS);
      }
      RSTAMP(1);

      // ---- team barrier n-1: plane n-1 is complete, slot n % 2 has been read ------------
      if (do_gather && wid == 0) {
        int fail = 0;
        if (lane == 0 && !RDEBUG(a, 4)) {
          const unsigned target = (unsigned)TEAM_WGS * (unsigned)n;
          unsigned spins = 0;
          unsigned long long t0 = 0;
          while (seen < target) {
            __builtin_amdgcn_s_sleep(1);
            seen = rflag_load(bar);
            if ((++spins & 255) == 0) {
              if (t0 == 0) t0 = __builtin_amdgcn_s_memrealtime();
              if (rflag_load(&ctl[RCTL_ERROR]) != 0) {
                fail = 1;
                break;
              }
              if (__builtin_amdgcn_s_memrealtime() - t0 > RING_TIMEOUT_TICKS) {
                rflag_store(&ctl[RCTL_ERROR], 1u);
                fail = 1;
                break;
              }
            }
          }
        }
        fail = rsgpr(fail);
        if (fail && lane == 0) s_info[2] = 1;
      }
      RING_WAIT_LDS();
      RING_BARRIER();
      if (rsgpr(s_info[2])) {
        failed = true;
        break;
      }
      RSTAMP(2);

      // every wave of the workgroup is past its MFMAs: the next tap's weights may land
      if (n + 1 < NP) stage_weights(t + 1 == T ? 0 : t + 1);

      // ---- plane n leaves for slot n % 2 -------------------------------------------------
      if (do_mfma && active && !RDEBUG(a, 1) && y_ok) {
        unsigned char *dst = ropaque_ptr(ring + (size_t)((n & 1) * plane_bytes));
        *(GANET_GLOBAL u32x4 *)(dst + y_off) = ypk[0];
        *(GANET_GLOBAL u32x4 *)(dst + y_off + 64) = ypk[1];
      }

      // ---- plane n-1 is gathered from the other slot and blended -------------------------
      if (do_gather && active && !RDEBUG(a, 2)) {
        const int so = (int)(((n - 1) & 1) * plane_bytes) + pj * 16;
        u32x4 v[2][4];
        float lh[2], lw[2], mk[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const uint4 e = tab_rd[st * 8 * TROW + tp];
          const int base = (int)(e.x & ~15u) + so;
          const unsigned c1 = (e.x & 1u) ? (unsigned)base : OOB_OFFSET;
          const unsigned c2 = (e.x & 2u) ? (unsigned)(base + PXB) : OOB_OFFSET;
          const unsigned c3 = (e.x & 4u) ? (unsigned)(base + row_bytes) : OOB_OFFSET;
          const unsigned c4 = (e.x & 8u) ? (unsigned)(base + row_bytes + PXB) : OOB_OFFSET;
          v[st][0] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c1, 0, AUX_SC1);
          v[st][1] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c2, 0, AUX_SC1);
          v[st][2] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c3, 0, AUX_SC1);
          v[st][3] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c4, 0, AUX_SC1);
          lh[st] = __uint_as_float(e.y);
          lw[st] = __uint_as_float(e.z);
          mk[st] = __uint_as_float(e.w);
        }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const float hh = 1.f - lh[st], hw_ = 1.f - lw[st];
          racc_piece<TA>(v[st][0], hh * hw_ * mk[st], oacc[st]);
          racc_piece<TA>(v[st][1], hh * lw[st] * mk[st], oacc[st]);
          racc_piece<TA>(v[st][2], lh[st] * hw_ * mk[st], oacc[st]);
          racc_piece<TA>(v[st][3], lh[st] * lw[st] * mk[st], oacc[st]);
        }
      }
      RSTAMP(3);

      // ---- an image is complete: its sums leave, the next image's coordinates are formed
      if (img_edge && active) {
        store_out(img - NTEAMS);
        zero_out();
        if (do_mfma) form_table(img);
      }
      RSTAMP(4);

      RING_WAIT_ALL();
      RING_BARRIER();
      if (tid == 0) rflag_add(bar, 1u);
      RSTAMP(5);
      if (++t == T) {
        t = 0;
        img += NTEAMS;
      }
    }
#ifdef GANET_RING_DEBUG
    if (lane == 0)
      for (int ph = 0; ph < 8; ++ph)
        __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long *)(ctl + RCTL_STAMPS) + ph,
                               rs_acc[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
  }

  // ---- the launch is complete when every workgroup has finished and every image was taken
  __syncthreads();
  if (tid == 0) {
    if (!failed && rank == 0 && nimg > 0) rflag_add(&ctl[RCTL_IMAGES], (unsigned)nimg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned f = rflag_add(&ctl[RCTL_FINISHED], 1u);
    if (f + 1u == (unsigned)a.grid) {
      if (rflag_load(&ctl[RCTL_IMAGES]) != (unsigned)a.B && rflag_load(&ctl[RCTL_ERROR]) == 0)
        rflag_store(&ctl[RCTL_ERROR], 4u);
    }
  }
}

int ring_env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

size_t ring_plane_bytes(const ganet_dcn_params *p) {
  return ganet_align_up((size_t)p->H * p->W * PXB, 256);
}

}  // namespace

// 16-bit arithmetic, Cin in {64, 128, 256} as one K chunk, 64 output channels per tap, "same"
// output size, at most 16 taps, a whole plane per team (H*W <= 4096 pixels)
bool ring_applicable(const ganet_dcn_params *p, int ta, int Cp, int CoutP, int Ho, int Wo) {
  if (ta == GANET_F32) return false;
  if (Cp != 64 && Cp != 128 && Cp != 256) return false;
  if (CoutP != 64 || p->deformable_groups != 1 || p->groups != 1) return false;
  if (p->sH != 1 || p->sW != 1 || Ho != p->H || Wo != p->W) return false;
  if (p->kH * p->kW > RING_TMAX) return false;
  return (long long)p->H * p->W <= TEAM_PX;
}

size_t ring_workspace_bytes(const ganet_dcn_params *p) {
  return RING_CTL_BYTES + (size_t)NTEAMS * NSLOT * ring_plane_bytes(p);
}

template <typename TA, typename TO>
static int ring_launch(const RingArgs &a, int Cp, hipStream_t s) {
  if (Cp == 256)
    hipLaunchKernelGGL((lfa_ring_kernel<TA, TO, 256>), dim3(a.grid), dim3(RNT), 0, s, a);
  else if (Cp == 128)
    hipLaunchKernelGGL((lfa_ring_kernel<TA, TO, 128>), dim3(a.grid), dim3(RNT), 0, s, a);
  else
    hipLaunchKernelGGL((lfa_ring_kernel<TA, TO, 64>), dim3(a.grid), dim3(RNT), 0, s, a);
  return ganet_check_launch();
}

// xh: NHWC map in 16-bit arithmetic dtype `ta` with Cp channels; wq: packed weights
// [T][64][Cp]; ws: ring_workspace_bytes() of scratch (control block first).
int ring_forward(const void *xh, const float *offset, const float *mask, const void *wq,
                 const void *bias, void *out, const ganet_dcn_params *p, int ta, int Cp,
                 void *ws, hipStream_t s) {
  RingArgs a;
  a.x = xh;
  a.wq = wq;
  a.ctl = static_cast<unsigned int *>(ws);
  a.ring = static_cast<unsigned char *>(ws) + RING_CTL_BYTES;
  a.offset = offset;
  a.mask = mask;
  a.bias = bias;
  a.out = out;
  a.B = p->B;
  a.H = p->H;
  a.W = p->W;
  a.Cout = p->Cout;
  a.T = p->kH * p->kW;
  a.kW = p->kW;
  a.pH = p->pH;
  a.pW = p->pW;
  a.dH = p->dH;
  a.dW = p->dW;
  a.out_nhwc = p->out_layout == GANET_NHWC;
  a.relu = p->fuse_relu;
  a.points = p->offset_is_points;
  a.plane_bytes = (unsigned)ring_plane_bytes(p);
  a.grid = NTEAMS * TEAM_WGS;
  a.debug = ring_env_int("GANET_RING_DEBUG_BITS", 0);
  // the schedule is written for 8 XCDs of 32 CUs
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return GANET_ERR_HIP;
  if (cus != RING_XCDS * TEAM_WGS) return GANET_ERR_UNSUPPORTED;
  if (hipMemsetAsync(ws, 0, RING_CTL_BYTES, s) != hipSuccess) return GANET_ERR_HIP;
  if (ta == GANET_BF16 && p->dtype == GANET_F32) return ring_launch<bf16_t, float>(a, Cp, s);
  if (ta == GANET_BF16) return ring_launch<bf16_t, bf16_t>(a, Cp, s);
  return ring_launch<f16_t, f16_t>(a, Cp, s);
}

}  // namespace ganet_dcn
