// Deformable convolution forward, contract-then-sample form with the per-tap products kept
// on chip ("algorithm 4") for gfx950 (MI355X).
//
// Algorithm 2 (csrc/deform_conv_cts.hip) replaces deformable_im2col + addmm_
// (mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:222-236) by  Y_t = X * W_t^T  and
// out = sum_t sample(Y_t), and pays for it with Y's round trip through the fabric: 918 of the
// 1272 MB one LFA call moves at 128 images.  Here Y never leaves the XCD that formed it:
//
//   * the 32 workgroups of an XCD (1024 threads each, one per CU) are a TEAM that owns two
//     images at a time (half 0: waves 0..255 of the team, half 1: waves 256..511); a wave
//     owns 16 pixels of its image and keeps their x (16 x Cin) in registers as MFMA operands
//     for the image's whole life;
//   * Y is formed TAP-MAJOR: in step m every wave multiplies its x block with W_t (32 KB,
//     staged in LDS two steps ahead as ready-made 16x16x32 operand images, double buffered)
//     and stores its 16 pixels of plane m (128 B per pixel) into slot m % 3 of the team's
//     ring: three planes of 2 * H*W*128 B, the same 3 MB of addresses for every tap and every
//     image - the intent was that the lines are dirtied again in L2 instead of being written
//     back; MEASURED: every stored byte leaves L2 all the same (WRITE_SIZE = all of Y), only
//     the way back is saved, and the launch is slower than algorithm 2
//     (profiles/r03_algo4_notes.txt): opt-in, kept as the record;
//   * in the same step the wave gathers plane m-2 (8 lanes x 16 B per pixel, a corner
//     outside the map reads 0 through the buffer bounds check) and adds it into the fp32 sums
//     of the 16 output pixels it owns; sums, x and coordinates stay in registers / LDS across
//     the T taps.
//
// Schedule: static.  A workgroup reads its XCD from the hardware register and draws a ticket
// from that XCD's counter: the ticket is its rank in the team; team x takes images x, 8 + x
// (first pair), 16 + x, 24 + x, ...  There are no queues and no claims.  Flow control is one
// monotonic counter per team in a cache line of its own.  A workgroup adds 1 in the middle of
// every step m, behind its MFMAs (ARRIVAL m), once every wave of it has seen the previous
// step's store complete - which a counted s_waitcnt tells without draining the gathers just
// sent, because a wave's vector memory operations complete in order.  Arrival m of every
// workgroup says that planes <= m-1 have reached L2 and that the gathers of planes <= m-3 have
// returned, which is what it takes to gather plane m-1 and to overwrite plane m-3's slot with
// plane m: one condition per step, checked at the END of step m for arrivals posted in its
// middle.  Nothing is drained inside the loop, and the only workgroup barrier of a step (the
// weights' double buffer and the arrival need one) stands between the MFMAs and the blend.
// Visibility: producer and consumer share one L2 by construction (teams are formed per
// XCC id), Y is stored by plain stores (L2 keeps the lines for its readers), every storing
// wave has seen them complete (the counted wait above) before its workgroup arrives, and every
// load of Y is an sc1 load
// (served by L2, never by the CU's L1, which may hold the slot's previous plane).
// Every wait is bounded; a wait that runs out leaves a non-zero status word, and the waves
// that see it stop touching memory and only keep their workgroup's barriers.  The last
// workgroup to finish checks that every image was taken by some team.
//
// Arithmetic of the sampling position and weights: deform_conv_cuda_kernel.cu:84-115,
// 218-229 (and :604-627 for the DCNv2 mask), unchanged; blend order as in algorithm 2, and
// the products are the same bits (fp32 accumulation over the channels in the same order).

#include "dcn_device.h"
#include <cstdlib>

namespace ganet_dcn {

namespace {

constexpr int RW = 16;                       // waves per workgroup, one workgroup per CU
constexpr int RNT = RW * 64;
constexpr int TEAM_WGS = 32;                 // workgroups per team = CUs per XCD
constexpr int RING_XCDS = 8;                 // teams
constexpr int HALVES = 2;                    // images a team works on at a time
constexpr int WPX = 16;                      // pixels per wave
constexpr int HALF_WAVES = TEAM_WGS * RW / HALVES;
constexpr int HALF_PX = HALF_WAVES * WPX;    // 4096 pixels per image
constexpr int NSLOT = 3;
constexpr int PXB = 128;                     // bytes per pixel of a plane (64 outputs)
constexpr int RING_TMAX = 16;                // taps
constexpr unsigned long long RING_TIMEOUT_TICKS = 200000000ull;  // 2 s of the 100 MHz clock
#ifndef GANET_RING_GATHER_AUX
#define GANET_RING_GATHER_AUX 16             /* sc1: served by L2, never by the CU's L1 */
#endif
constexpr int AUX_SC1 = GANET_RING_GATHER_AUX;
#ifndef GANET_RING_STORE_MOD
#define GANET_RING_STORE_MOD ""              /* cache policy of the Y stores (tuning builds: " nt", ...) */
#endif

// control block (32-bit words); hot words sit in 128-byte lines of their own
enum {
  RL = 32,
  RCTL_ERROR = 0,
  RCTL_FINISHED = 1 * RL,
  RCTL_IMAGES = 2 * RL,
  RCTL_JOIN = 3 * RL,                        // [8] one line per XCC id
  RCTL_BAR = RCTL_JOIN + RING_XCDS * RL,     // [8] one line per team
  RCTL_WORDS = RCTL_BAR + RING_XCDS * RL
};
constexpr size_t RING_CTL_BYTES = 4096;
static_assert(RCTL_WORDS * 4 <= RING_CTL_BYTES, "control block");

struct RingArgs {
  const void *x;        // [B][H*W][KH] channel-last, 16-bit arithmetic dtype
  const void *wq;       // [T][64][KH]
  unsigned char *ring;  // [RING_XCDS][NSLOT][HALVES][plane_bytes]
  const float *offset;  // [B][2T][H][W]
  const float *mask;    // [B][T][H][W] or nullptr
  const void *bias;     // [Cout] or nullptr
  void *out;
  unsigned int *ctl;
  int B, H, W, Cout, T;
  int kW, pH, pW, dH, dW;
  int out_nhwc, relu, points;
  unsigned plane_bytes;
  int grid;             // workgroups launched (RING_XCDS * TEAM_WGS)
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

// Timing experiments, compiled in by -DGANET_RING_ABLATE=<bits> (a separate library, see
// ganet_amd/build.py): 1 nothing produced (no MFMA, no x load, no Y store), 2 nothing sampled,
// 4 no waiting on the team's counter, 8 no weight staging, 16 no image-edge work (sums not
// stored, coordinates and x of the following images not fetched).  Results are then undefined.
#ifndef GANET_RING_ABLATE
#define GANET_RING_ABLATE 0
#endif
#define RABL(bit) (GANET_RING_ABLATE & (bit))

// the wave's s_waitcnt forms the compiler's own wait bookkeeping understands (gfx9 encoding:
// vmcnt[3:0] | expcnt << 4 | lgkmcnt << 8 | vmcnt[5:4] << 14)
#define RING_WAIT_ALL() __builtin_amdgcn_s_waitcnt(0x0070)        /* vmcnt(0) lgkmcnt(0) */
#define RING_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f)        /* lgkmcnt(0) */
#define RING_BARRIER() asm volatile("s_barrier" ::: "memory")

// Vector memory operations the compiler's wait bookkeeping must not see: beside a pending
// store or atomic it stops counting (stores and loads are different event kinds to it, so it
// waits for vmcnt(0) ahead of every use of a loaded register), and beside an LDS-DMA in flight it
// waits for vmcnt(0) ahead of every LDS read.  The hardware completes them in issue order with the
// loads, which the waits written out in the step loop rely on; an unseen operation only makes
// the compiler's own counted waits wait for more.  None of them has a register result.
__device__ __forceinline__ void asm_store128(const void *sbase, unsigned voff, const u32x4 &v) {
  asm volatile("global_store_dwordx4 %0, %1, %2" GANET_RING_STORE_MOD ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void asm_store128_64(const void *sbase, unsigned voff, const u32x4 &v) {
  asm volatile("global_store_dwordx4 %0, %1, %2 offset:64" GANET_RING_STORE_MOD ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}
__device__ __forceinline__ void asm_atomic_inc(const void *sbase) {
  const unsigned zero = 0u, one = 1u;
  asm volatile("global_atomic_add %0, %1, %2" ::"v"(zero), "v"(one), "s"(sbase) : "memory");
}
// LDS-DMA of 16 bytes per lane: LDS destination = lds_dst (wave-uniform byte address) + 16 * lane
__device__ __forceinline__ void asm_glds128(const void *sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_dst)
      : "memory");
}

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
  // one array: two weight buffers, the waves' coordinate tables, a few words of workgroup state
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * W_BYTES + TAB_BYTES + 64];
  int *s_info = reinterpret_cast<int *>(smem + 2 * W_BYTES + TAB_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = rsgpr(tid >> 6);
  unsigned int *ctl = a.ctl;

  // ---- join: XCD = team, rank ------------------------------------------------------------
  if (tid == 0) {
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    xid &= 7u;
    const unsigned ticket = rflag_add(&ctl[RCTL_JOIN + xid * RL], 1u);
    s_info[0] = (int)xid;
    s_info[1] = (int)ticket;
  }
  __syncthreads();
  const int xcc = rsgpr(s_info[0]);
  const int rank = rsgpr(s_info[1]);
  const bool seated = rank < TEAM_WGS;       // else: more workgroups on this XCD than the schedule has
  const int HW = a.H * a.W;
  const int T = a.T;
  // images of this team: xcc, xcc + 8, ...; two at a time
  const int nimg = (!seated || xcc >= a.B) ? 0 : (a.B - xcc + RING_XCDS - 1) / RING_XCDS;
  const int npairs = (nimg + HALVES - 1) / HALVES;
  const int NP = rsgpr(npairs * T);
  if (!seated && tid == 0) rflag_store(&ctl[RCTL_ERROR], 2u);

  if (NP > 0) {
    unsigned int *bar = rsgpr_ptr(ctl + RCTL_BAR + xcc * RL);
    // pixels: the wave's 16 as MFMA columns (lane & 15) and as two gather groups of 8
    const int wg = rank * RW + wid;          // wave of the team
    const int half = rsgpr(wg / HALF_WAVES);
    const int px0 = rsgpr((wg - half * HALF_WAVES) * WPX);
    const bool has_px = px0 < HW;            // wave-uniform
    const int mc = lane & 15, mg = lane >> 4;  // MFMA: column (pixel), k group / row group
    const int pxg = lane >> 3, pj = lane & 7;  // gather: pixel of the group, 16-byte piece
    const unsigned plane_bytes = a.plane_bytes;
    const unsigned slot_bytes = HALVES * plane_bytes;
    unsigned char *ring = rsgpr_ptr(a.ring + (size_t)xcc * NSLOT * slot_bytes + (size_t)half * plane_bytes);
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
        ring, 0, (int)(NSLOT * slot_bytes - half * plane_bytes), 0x00020000);
    const int row_bytes = a.W * PXB;
    const uint4 *tab_rd = reinterpret_cast<const uint4 *>(smem + 2 * W_BYTES) + wid * (WPX * TROW) + pxg * TROW;
    uint4 *tab_wr = reinterpret_cast<uint4 *>(smem + 2 * W_BYTES) + wid * (WPX * TROW) + pxg * TROW;
    const unsigned char *xbytes = static_cast<const unsigned char *>(a.x);
    const unsigned char *wq = static_cast<const unsigned char *>(a.wq);
    const size_t img_x_bytes = (size_t)HW * KH * 2;
    const int B = a.B;

    // per-lane byte offsets (32 bit) against wave-uniform bases
    const unsigned y_off = (unsigned)(px0 + mc) * PXB + (unsigned)mg * 16u;
    const bool y_ok = px0 + mc < HW;
    // weights of tap t into LDS as operand images: image (m, ks) holds, for lane l, the 16
    // bytes W_t[row(m, l & 15)][32 ks + 8 (l >> 4) ..+8].  Row r = 4 g + i of tile m is output
    // 32 (m >> 1) + 8 g + 4 (m & 1) + i: accumulator register i of lane (pixel, g) in tiles
    // 2h, 2h + 1 then holds outputs 32 h + 8 g .. +8 of that pixel, 16 contiguous bytes of Y.
    const unsigned w_off = (unsigned)((8 * ((lane & 15) >> 2) + (lane & 3)) * KH + (lane >> 4) * 8) * 2u;
    const unsigned lds_base = (unsigned)(size_t)(lds_void *)smem;   // LDS byte address of the array
    auto stage_weights = [&](int t, int buf) {
      t = ropaque(t);
#pragma unroll
      for (int j = 0; j < NIMG / RW; ++j) {
        const int ii = j * RW + wid;
        const int m = ii / NKS, ks = ii - m * NKS;
        const unsigned char *base = wq + ((size_t)(t * 64 + 32 * (m >> 1) + 4 * (m & 1)) * KH + ks * 32) * 2;
        asm_glds128(base, w_off, lds_base + (unsigned)(buf * W_BYTES + ii * 1024));
      }
    };

    u32x4 xb[NKS];
    auto load_x = [&](int img) {
      img = ropaque(img);
      const unsigned char *base = ropaque_ptr(xbytes + (size_t)img * img_x_bytes);
      const int pxo = ropaque(px0), hw = ropaque(HW);   // (formed here, once per image, not kept)
      const int pm = pxo + mc < hw ? pxo + mc : hw - 1;
      const unsigned x_off = (unsigned)pm * (KH * 2) + (unsigned)mg * 16u;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks) xb[ks] = ldg128(base + x_off + ks * 64);
    };

    // table of image img: lane (pxg, pj) forms taps pj, pj + 8 of pixels pxg and 8 + pxg.
    // The image-edge paths read their arguments again from the kernel argument segment (scalar
    // loads, once per image) instead of holding them in SGPRs through the step loop.
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

    // image of pair i for this wave's half, and whether the wave has work on it
    auto image_of = [&](int pair) { return (pair * HALVES + half) * RING_XCDS + xcc; };

    // ---- the team's counter: `seen` is the last value this wave has read (monotonic) -------
    unsigned seen = 0;
    bool dead = false;                       // a wait ran out somewhere: only barriers from now on
    auto wait_count = [&](unsigned target) {
      if (seen >= target || dead || RABL(4)) return;
      int st = 0;
      if (lane == 0) {
        unsigned s = rflag_load(bar);
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (s < target) {
          __builtin_amdgcn_s_sleep(1);
          s = rflag_load(bar);
          if ((++spins & 255) == 0) {
            if (t0 == 0) t0 = __builtin_amdgcn_s_memrealtime();
            if (rflag_load(&ctl[RCTL_ERROR]) != 0) {
              st = 1;
              break;
            }
            if (__builtin_amdgcn_s_memrealtime() - t0 > RING_TIMEOUT_TICKS) {
              rflag_store(&ctl[RCTL_ERROR], 1u);
              st = 1;
              break;
            }
          }
        }
        seen = s;
      }
      seen = (unsigned)rsgpr((int)seen);
      if (rsgpr(st)) dead = true;
    };

    // ---- prologue: the first two taps' weights, the first image's x and coordinates -------
    if (!RABL(8)) {
      stage_weights(0, 0);
      if (NP > 1) stage_weights(1 % T, 1);
    }
    if (has_px && image_of(0) < B) {
      if (!RABL(1)) load_x(image_of(0));
      form_table(image_of(0));
    }
    zero_out();
    RING_WAIT_ALL();
    RING_BARRIER();
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(xb[ks]));   // (x is complete: see step 5)

    // Step m of a wave (planes m and m-2; no memory operation is ever drained in the loop, the
    // in-order completion of a wave's vector memory operations carries the hand-offs):
    //   1  the gathers of plane m-2 go out (8 loads of 1 KB per wave)
    //   2  plane m is formed on the matrix cores while they are in flight
    //   3  vmcnt(8): everything this wave sent before those gathers is complete - the store of
    //      plane m-1, the weights it staged in step m-1 - then the workgroup's barrier; one lane
    //      adds the workgroup to the team's counter (ARRIVAL m: planes <= m-1 are stored, the
    //      gathers of planes <= m-3 have returned); plane m+2's weights go to the buffer plane
    //      m's sat in (LDS-DMA); a look at the counter is sent
    //   4  plane m-2 is blended into the sums; at an image's last tap the sums leave and the
    //      next image's coordinates are formed
    //   5  once the counter shows arrival m of every workgroup - posted half a step ago - plane
    //      m is stored into slot m % 3 (it held plane m-3) and step m+1 may gather plane m-1;
    //      where plane m was an image's last, the next image's x is fetched
    u32x4 ypk[2];
    int t = 0, pair = 0;                     // tap and pair of plane m
    int tq = 0, pq = 0;                      // ... of plane m-1 (read from m = 1 on)
    int tg = 0, pg = 0;                      // ... of plane m-2 (read from m = 2 on)
    for (int m = 0; m <= NP + 1; ++m) {
      const bool has_m = m < NP;             // plane m exists
      const bool has_g = m >= 2;             // plane m-2 exists (m - 2 < NP always holds here)
      const bool act_m = has_m && has_px && image_of(pair) < B && !dead && !RABL(1);
      const bool act_g = has_g && has_px && image_of(pg) < B && !dead;

      // ---- 1: gathers of plane m-2 (the counter was checked at the end of step m-1) --------
      // Every wave sends its 8 loads in every step - a wave without work sends them out of
      // bounds, where they cost nothing and read 0 - so that the step is one straight line of
      // vector memory operations and every wait in it is a counted one.
      u32x4 v[2][4];
      if (!RABL(2)) {
        const int so = (int)((unsigned)((m + NSLOT - 2) % NSLOT) * slot_bytes) + pj * 16;
        const unsigned off_all = act_g ? 0u : OOB_OFFSET;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          const uint4 e = tab_rd[st * 8 * TROW + tg];
          const int base = (int)(e.x & ~15u) + so;
          const unsigned c1 = ((e.x & 1u) ? (unsigned)base : OOB_OFFSET) | off_all;
          const unsigned c2 = ((e.x & 2u) ? (unsigned)(base + PXB) : OOB_OFFSET) | off_all;
          const unsigned c3 = ((e.x & 4u) ? (unsigned)(base + row_bytes) : OOB_OFFSET) | off_all;
          const unsigned c4 = ((e.x & 8u) ? (unsigned)(base + row_bytes + PXB) : OOB_OFFSET) | off_all;
          v[st][0] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c1, 0, AUX_SC1);
          v[st][1] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c2, 0, AUX_SC1);
          v[st][2] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c3, 0, AUX_SC1);
          v[st][3] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, c4, 0, AUX_SC1);
        }
      }

      // ---- 2: plane m = x_block * W_t^T on the matrix cores (weights in buffer m & 1) ------
      if (act_m) {
        // One 16x16 tile of outputs at a time (a chain of NKS MFMAs on one accumulator runs at
        // the pipe's rate), the weight fragments read PF MFMAs ahead.  The order is pinned: left
        // to itself the scheduler, short of registers here, reads each fragment right before
        // its MFMA and every MFMA then waits out an LDS round trip.
        const u32x4 *wimg = reinterpret_cast<const u32x4 *>(smem + (m & 1) * W_BYTES) + lane;
        constexpr int NF = 4 * NKS, PF = 3;
        frag_t fr[PF];
#pragma unroll
        for (int i = 0; i < PF; ++i) fr[i] = __builtin_bit_cast(frag_t, wimg[i * 64]);
        TA o[8];
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          const int q = i / NKS, ks = i % NKS;
          const frag_t fa = fr[i % PF];
          __builtin_amdgcn_sched_barrier(0);
          acc = Mma16<TA>::mma(fa, __builtin_bit_cast(frag_t, xb[ks]), acc);
          if (i + PF < NF) fr[i % PF] = __builtin_bit_cast(frag_t, wimg[(i + PF) * 64]);
          __builtin_amdgcn_sched_barrier(0);
          if (ks == NKS - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[4 * (q & 1) + e] = from_f32<TA>(acc[e]);
            acc = f32x4{0.f, 0.f, 0.f, 0.f};
            if (q & 1) __builtin_memcpy(&ypk[q >> 1], o, 16);
          }
        }
      }

      // ---- 3: step m-1 of this wave is complete; barrier; arrival m; weights; x -----------
      if (!RABL(2))
        __builtin_amdgcn_s_waitcnt(0x0078);   // vmcnt(8) lgkmcnt(0): all but the 8 gathers
      else
        RING_WAIT_ALL();
      RING_BARRIER();
      if (tid == 0 && m >= 1) asm_atomic_inc(bar);
      // every wave is past plane m's MFMAs: plane m+2's weights take the buffer (they have until
      // the next step's barrier to land)
      if (has_m && m + 2 < NP && !RABL(8)) stage_weights((t + 2) % T, m & 1);
      // a look at the counter, sent now and read at the end of the step: by then it has come
      // back, and it usually shows every workgroup's arrival m already
      unsigned pv = 0;
      if (lane == 0 && !dead && !RABL(4)) pv = rflag_load(bar);

      // ---- 4: blend plane m-2; an image's last tap closes it ------------------------------
      if (!RABL(2)) {
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          // (the loaded rows are first looked at here: without this the scheduler moves their
          // unpacking up between the MFMAs, and the barrier then waits for the gathers)
          asm volatile("" : "+v"(v[st][0]), "+v"(v[st][1]), "+v"(v[st][2]), "+v"(v[st][3]));
          // (the entry is read again rather than kept in registers across the MFMAs)
          const uint4 e = tab_rd[st * 8 * TROW + tg];
          const float lh = act_g ? __uint_as_float(e.y) : 0.f;
          const float lw = act_g ? __uint_as_float(e.z) : 0.f;
          const float mk = act_g ? __uint_as_float(e.w) : 0.f;
          const float hh = 1.f - lh, hw_ = 1.f - lw;
          racc_piece<TA>(v[st][0], hh * hw_ * mk, oacc[st]);
          racc_piece<TA>(v[st][1], hh * lw * mk, oacc[st]);
          racc_piece<TA>(v[st][2], lh * hw_ * mk, oacc[st]);
          racc_piece<TA>(v[st][3], lh * lw * mk, oacc[st]);
        }
      }
      if (has_g && tg == T - 1 && !RABL(16)) {
        if (act_g) {
          store_out(image_of(pg));
          zero_out();
          RING_WAIT_ALL();   // (its stores are the only ones the compiler sees: none pending at the join)
        }
        if (has_px && !dead && pg + 1 < npairs && image_of(pg + 1) < B) form_table(image_of(pg + 1));
      }

      // ---- 5: arrival m of every workgroup; plane m leaves for slot m % 3 ------------------
      pv = (unsigned)rsgpr((int)pv);
      if (pv > seen) seen = pv;
      if (m >= 1 && m <= NP) wait_count((unsigned)TEAM_WGS * (unsigned)m);
      if (act_m && !dead && y_ok) {
        const unsigned char *dst = ring + (size_t)(m % NSLOT) * slot_bytes;
        asm_store128(dst, y_off, ypk[0]);
        asm_store128_64(dst, y_off, ypk[1]);
      }
      // plane m was an image's last: the next image's x takes the registers.  Fetch and wait in
      // ONE block: the compiler's wait bookkeeping merges paths, and a fetch in one conditional
      // block with its first use in another makes it wait for every load ahead of every step's
      // MFMAs, the gathers just sent included.
      if (act_m && t == T - 1 && image_of(pair + 1) < B && !RABL(16)) {
        load_x(image_of(pair + 1));
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) asm volatile("" : "+v"(xb[ks]));
      }

      // advance the three (tap, pair) clocks
      tg = tq; pg = pq;
      tq = t; pq = pair;
      if (++t == T) {
        t = 0;
        ++pair;
      }
    }
  }

  // ---- the launch is complete when every workgroup has finished and every image was taken
  __syncthreads();
  if (tid == 0) {
    if (seated && rank == 0 && nimg > 0) rflag_add(&ctl[RCTL_IMAGES], (unsigned)nimg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned f = rflag_add(&ctl[RCTL_FINISHED], 1u);
    if (f + 1u == (unsigned)a.grid) {
      if (rflag_load(&ctl[RCTL_IMAGES]) != (unsigned)a.B && rflag_load(&ctl[RCTL_ERROR]) == 0)
        rflag_store(&ctl[RCTL_ERROR], 4u);
    }
  }
}

size_t ring_plane_bytes(const ganet_dcn_params *p) {
  return ganet_align_up((size_t)p->H * p->W * PXB, 256);
}

}  // namespace

// 16-bit arithmetic, Cin in {64, 128, 256} as one K chunk, 64 output channels per tap, "same"
// output size, at most 16 taps, a whole plane per image half of a team (H*W <= 4096 pixels)
bool ring_applicable(const ganet_dcn_params *p, int ta, int Cp, int CoutP, int Ho, int Wo) {
  if (ta == GANET_F32) return false;
  if (Cp != 64 && Cp != 128 && Cp != 256) return false;
  if (CoutP != 64 || p->deformable_groups != 1 || p->groups != 1) return false;
  if (p->sH != 1 || p->sW != 1 || Ho != p->H || Wo != p->W) return false;
  if (p->kH * p->kW > RING_TMAX) return false;
  return (long long)p->H * p->W <= HALF_PX;
}

size_t ring_workspace_bytes(const ganet_dcn_params *p) {
  return RING_CTL_BYTES + (size_t)RING_XCDS * NSLOT * HALVES * ring_plane_bytes(p);
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
  a.grid = RING_XCDS * TEAM_WGS;
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
