// Deformable convolution forward, "contract-then-sample" form, for gfx950 (MI355X).
//
// Bilinear sampling and the channel contraction are both linear, so they commute:
//
//   out[p][o] = sum_t sum_c W[o][c][t] * sample(x[:, c], pos(p, t))          (reference order:
//                 deformable_im2col + addmm_, deform_conv_cuda.cpp:222-236)
//             = sum_t sample(Y_t[:, o], pos(p, t)),   Y_t[q][o] = sum_c x[q][c] * W[o][c][t]
//
// When Cout < Cin (the LFA: 256 -> 64) the second form blends 4 corners x Cout values
// per (pixel, tap) instead of 4 x Cin, and gathers Cout-wide rows instead of Cin-wide
// ones: 4x less VALU and 4x less L2->L1 gather traffic, which is what bounds the fused
// sample-then-contract kernel (profiles/r01_ablation.txt).  The price is the
// intermediate Y[pixel][tap][Cout]; the host loop walks the batch in chunks whose Y
// stays resident in the 256 MiB Infinity Cache between the two kernels, so HBM still
// sees x, the offsets and the output once.
//
//   cts_gemm_kernel    Y = X * Wq^T  on the matrix cores: every wavefront keeps the A
//                      fragments of its 32 pixels in registers and walks the N = T*Cout
//                      axis in 64-column slices; the weight slice is double-buffered in LDS.
//   cts_sample_kernel  per output pixel: coordinates once per tap (the lanes of a pixel
//                      split the taps and share them through LDS), 4 buffer loads of
//                      16 B per lane and tap (a corner outside the map reads 0 by the
//                      buffer bounds check), fp32 blend/accumulate, bias, ReLU, store.
//
// Arithmetic of the sampling position and weights: deform_conv_cuda_kernel.cu:84-115,
// 218-229 (and :604-627 for the DCNv2 mask), unchanged.

#include "dcn_device.h"
#include <cstdlib>

namespace ganet_dcn {

namespace {

#ifndef GANET_GEMM_WAVES
#define GANET_GEMM_WAVES 8
#endif
constexpr int GEMM_WAVES = GANET_GEMM_WAVES;
constexpr int SAMPLE_WAVES = 4;
constexpr int TAP_BLOCK = 16;   // taps whose coordinates sit in LDS at a time

// ---- Y = X * Wq^T -------------------------------------------------------------
// x [M][Cp], wq [N][Cp] (N multiple of 64), y [M][N].
//
// A wavefront owns RB blocks of 32 rows (pixels) and keeps their KH channels in
// registers as MFMA operands for its whole life; the workgroup walks the N axis in
// 64-column slices whose weights are double-buffered in LDS (one barrier per slice).
// Every weight fragment read from LDS feeds RB MFMAs: with RB = 1 the ds_read traffic
// alone takes as long as the MFMAs.  The product is formed transposed (rows = output
// columns, columns = pixels), so a lane ends up with 4 consecutive outputs of ONE pixel
// per accumulator quad: it parks them in the wave's LDS rows with 8/16-byte writes
// and whole row segments leave as 16-byte stores.
// One 8-wave workgroup per CU (256 VGPRs); two 4-wave workgroups per CU measured the
// same.  The next weight slice is staged through registers in two halves.
template <typename TA, typename TY, int KH, int RB>
__global__ __launch_bounds__(GEMM_WAVES * 64, 2) void cts_gemm_kernel(
    const TA *__restrict__ x, const TA *__restrict__ wq, TY *__restrict__ y,
    long long M, int Cp, int N, int slices_per_block, int ablate) {
  constexpr int ES = sizeof(TA);
  constexpr int EPL = 16 / ES;
  constexpr int NQ = KH / (2 * EPL);          // 16-byte fragments per lane and K chunk
  constexpr int LDB = KH + EPL;               // +16 B: conflict-free ds_read_b128
  constexpr int B_BYTES = 64 * LDB * ES;
  constexpr int NT = GEMM_WAVES * 64;
  constexpr int YROW = 32 * sizeof(TY) + 16;  // staged half-slice row (bytes)
  typedef typename Mma<TA>::frag frag_t;
  __shared__ __attribute__((aligned(16))) unsigned char s_b[2 * B_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_y[GEMM_WAVES * 32 * YROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long m0 = ((long long)blockIdx.x * GEMM_WAVES + wid) * (32 * RB);
  const TA *xrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const long long r = m0 + rb * 32 + i;
    xrow[rb] = x + (r < M ? r : M - 1) * Cp + h * EPL;
  }

  // blockIdx.y takes a range of the 64-column slices (small batches: more workgroups
  // than row tiles alone would give; x is then re-read from L2 once per range)
  const int kchunks = Cp / KH;
  const int slice0 = blockIdx.y * slices_per_block;
  const int nslices = min(slices_per_block, N / 64 - slice0);
  const int total = nslices * kchunks;
  wq += (size_t)slice0 * 64 * Cp;
  y += (size_t)slice0 * 64;

  constexpr int BPIECES = 64 * (KH / EPL);
  constexpr int BPT = BPIECES / NT;           // 16-byte pieces per thread and slice
  constexpr int NH = BPT % 2 == 0 ? 2 : 1;    // staged in two halves when it splits
  constexpr int BH = BPT / NH;
  static_assert(BPIECES % NT == 0, "weight slice must split evenly over the threads");
  u32x4 breg[BH];
  auto load_b = [&](int it, int half) {
    if (half >= NH) return;
    const int n = it / kchunks, kc = it - n * kchunks;
#pragma unroll
    for (int j = 0; j < BH; ++j) {
      const int q = tid + (half * BH + j) * NT;
      const int r = q / (KH / EPL), pc = q - r * (KH / EPL);
      breg[j] = ldg128(wq + ((size_t)(n * 64 + r) * Cp + kc * KH + pc * EPL));
    }
  };
  auto store_b = [&](int buf, int half) {
    if (half >= NH) return;
    TA *Bs = reinterpret_cast<TA *>(s_b + buf * B_BYTES);
#pragma unroll
    for (int j = 0; j < BH; ++j) {
      const int q = tid + (half * BH + j) * NT;
      const int r = q / (KH / EPL), pc = q - r * (KH / EPL);
      *reinterpret_cast<u32x4 *>(Bs + r * LDB + pc * EPL) = breg[j];
    }
  };

  frag_t fa[RB][NQ];
  auto load_a = [&](int kc) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        fa[rb][q] = __builtin_bit_cast(frag_t, ldg128(xrow[rb] + kc * KH + q * 2 * EPL));
  };

#ifdef GANET_ABLATE
  unsigned long long ts[40];
  int nts = 0;
  const bool stamp = (ablate & 32) && blockIdx.x == gridDim.x / 2 && tid == 0;
#define GANET_STAMP() do { if (stamp && nts < 40) ts[nts++] = wall_clock64(); } while (0)
#else
#define GANET_STAMP() do {} while (0)
#endif
  GANET_STAMP();
  load_b(0, 0);
  load_a(0);
  store_b(0, 0);
  load_b(0, 1);
  store_b(0, 1);
#ifdef GANET_ABLATE
  if (ablate & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  GANET_STAMP();

  f32x16 acc[RB][2];
  for (int it = 0; it < total; ++it) {
    const int n = it / kchunks, kc = it - n * kchunks;
    const bool more = it + 1 < total;
    __syncthreads();  // slice `it` is in its buffer; the other buffer is free again
    GANET_STAMP();
    if (more) load_b(it + 1, 0);
    if (kchunks > 1 && it > 0) load_a(kc);
    if (kc == 0) {
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc[rb][0][r] = 0.f;
          acc[rb][1][r] = 0.f;
        }
    }
    {
      const TA *Bs = reinterpret_cast<const TA *>(s_b + (it & 1) * B_BYTES);
      const TA *brow0 = Bs + i * LDB + h * EPL;
      const TA *brow1 = brow0 + 32 * LDB;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q == NQ / 2 && more) {   // first half has landed behind NQ/2 MFMA steps
          store_b((it + 1) & 1, 0);
          load_b(it + 1, 1);
        }
        const frag_t fb0 = *reinterpret_cast<const frag_t *>(brow0 + q * 2 * EPL);
        const frag_t fb1 = *reinterpret_cast<const frag_t *>(brow1 + q * 2 * EPL);
#ifdef GANET_ABLATE  // profiling build only: bit0 skips the MFMAs, operands stay live
        if (ablate & 1) {
          asm volatile("" ::"v"(fa[0][q]), "v"(fb0), "v"(fb1));
          continue;
        }
#endif
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          Mma<TA>::mma(fb0, fa[rb][q], acc[rb][0]);   // rows = outputs, columns = pixels
          Mma<TA>::mma(fb1, fa[rb][q], acc[rb][1]);
        }
      }
    }
    GANET_STAMP();
    if (more) store_b((it + 1) & 1, 1);
    GANET_STAMP();
    if (kc == kchunks - 1) {
#ifdef GANET_ABLATE  // bit2: keep only the first slice's stores
      if ((ablate & 4) && n > 0) {
        asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[0][1][0]));
        continue;
      }
#endif
      // lane (i, h), registers 4g..4g+3 of tile t hold outputs t*32 + 8g + 4h + 0..3 of
      // pixel i
      unsigned char *ys = s_y + wid * 32 * YROW;
      constexpr int LPR = 32 * sizeof(TY) / 16;   // lanes per staged row
      constexpr int RPP = 64 / LPR;               // rows per pass
      const int pc = lane % LPR, rq = lane / LPR;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          TY *yr = reinterpret_cast<TY *>(ys + i * YROW) + 4 * h;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            TY v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = from_f32<TY>(acc[rb][t][4 * g + e]);
            if constexpr (sizeof(TY) == 2) {
              unsigned long long pk;
              __builtin_memcpy(&pk, v, 8);
              *reinterpret_cast<unsigned long long *>(yr + 8 * g) = pk;
            } else {
              u32x4 pk;
              __builtin_memcpy(&pk, v, 16);
              *reinterpret_cast<u32x4 *>(yr + 8 * g) = pk;
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
          for (int ps = 0; ps < 32 / RPP; ++ps) {
            const int rr = ps * RPP + rq;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(ys + rr * YROW + pc * 16);
            const long long mr = m0 + rb * 32 + rr;
            if (mr < M)
              *reinterpret_cast<u32x4 *>(reinterpret_cast<unsigned char *>(
                  y + mr * N + (size_t)n * 64 + t * 32) + pc * 16) = v;
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    GANET_STAMP();
  }
#ifdef GANET_ABLATE
  if (stamp) {
    printf("gemm stamps (10 ns units, block %d):", (int)blockIdx.x);
    for (int k = 1; k < nts; ++k) printf(" %d", (int)(ts[k] - ts[0]));
    printf("\n");
  }
#endif
#undef GANET_STAMP
}

// ---- Y = X * Wq^T, 16-bit operands, results leaving under the next slice's MFMAs -----
// Same ownership as cts_gemm_kernel (a wavefront keeps the channels of RB x 32 pixels in
// registers) with 32-column slices and two alternating accumulator sets: while slice n is
// contracted, the finished 32x32 blocks of slice n-1 are converted, parked in the wave's
// LDS rows and stored - the staging writes, row reads and global stores sit between the
// MFMAs instead of behind them (in cts_gemm_kernel a slice costs 2.0 us of MFMAs plus
// 1.9 us of staging, stores and barrier skew in sequence: profiles/r01_ablation.txt).
// Stores are branch-free buffer stores (rows past M fall outside the window), so the
// wait for the next weight slice does not wait for them.
template <typename TA, typename TY, int KH, int RB>
__global__ __launch_bounds__(GEMM_WAVES * 64, 2) void cts_gemm_ov_kernel(
    const TA *__restrict__ x, const TA *__restrict__ wq, TY *__restrict__ y,
    long long M, int Cp, int N, int slices_per_block) {
  static_assert(sizeof(TA) == 2 && sizeof(TY) == 2, "16-bit operands and results");
  constexpr int EPL = 8;
  constexpr int NQ = KH / 16;
  constexpr int LDB = KH + EPL;               // +16 B: conflict-free ds_read_b128
  constexpr int SB = 32;                      // output columns per slice
  constexpr int B_BYTES = SB * LDB * 2;
  constexpr int NT = GEMM_WAVES * 64;
  constexpr int YROW = 2 * SB * 2 + 16;       // staged row of two slices' 32x32 blocks (bytes)
  static_assert(NQ >= 2 * RB, "one epilogue step per K fragment");
  typedef typename Mma<TA>::frag frag_t;
  __shared__ __attribute__((aligned(16))) unsigned char s_b[2 * B_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_y[GEMM_WAVES * RB * 32 * YROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long m0 = ((long long)blockIdx.x * GEMM_WAVES + wid) * (32 * RB);
  const TA *xrow[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const long long r = m0 + rb * 32 + i;
    xrow[rb] = x + (r < M ? r : M - 1) * Cp + h * EPL;
  }

  // Cp == KH (one K chunk: the host picks cts_gemm_kernel otherwise)
  const int slice0 = blockIdx.y * slices_per_block;
  const int nslices = min(slices_per_block, N / SB - slice0);
  if (nslices <= 0 || (nslices & 1)) return;
  wq += (size_t)slice0 * SB * Cp;
  y += (size_t)slice0 * SB;

  constexpr int BPIECES = SB * (KH / EPL);    // 16-byte pieces per slice
  constexpr int BPT = (BPIECES + NT - 1) / NT;
  u32x4 breg[BPT];
  auto load_b = [&](int n) {
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NT;
      if (BPIECES % NT != 0 && q >= BPIECES) break;
      const int r = q / (KH / EPL), pc = q - r * (KH / EPL);
      breg[j] = ldg128(wq + ((size_t)(n * SB + r) * Cp + pc * EPL));
    }
  };
  auto store_b = [&](int buf) {
    TA *Bs = reinterpret_cast<TA *>(s_b + buf * B_BYTES);
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NT;
      if (BPIECES % NT != 0 && q >= BPIECES) break;
      const int r = q / (KH / EPL), pc = q - r * (KH / EPL);
      *reinterpret_cast<u32x4 *>(Bs + r * LDB + pc * EPL) = breg[j];
    }
  };

  frag_t fa[RB][NQ];
  auto load_a = [&]() {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        fa[rb][q] = __builtin_bit_cast(frag_t, ldg128(xrow[rb] + q * 2 * EPL));
  };

  unsigned char *ys = s_y + wid * (RB * 32 * YROW);
  const __amdgpu_buffer_rsrc_t ydst = __builtin_amdgcn_make_buffer_rsrc(
      y, 0, (int)(((size_t)M * N - (size_t)slice0 * SB) * 2), 0x00020000);
  // Two slices share a staged row: the even slice parks its 32 columns in the left half,
  // the odd one in the right half, and the row leaves as one whole 128-byte line (halves
  // stored a slice apart do not merge in L2: WRITE_SIZE 136 MB against 115 MB of Y).
  // lane (i, h), registers 4g..4g+3 hold outputs 8g + 4h + 0..3 of pixel i
  auto stage_write = [&](const f32x16 &acc, int rb, int half) {
    TY *yr = reinterpret_cast<TY *>(ys + (rb * 32 + i) * YROW + half * (SB * 2)) + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      TY v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = from_f32<TY>(acc[4 * g + e]);
      unsigned long long pk;
      __builtin_memcpy(&pk, v, 8);
      *reinterpret_cast<unsigned long long *>(yr + 8 * g) = pk;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // rows of slices n_even and n_even + 1
  auto stage_store = [&](int n_even, int rb) {
    const int pc = lane & 7, rq = lane >> 3;   // 8 lanes x 16 B per 64-column row
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rr = ps * 8 + rq;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(ys + (rb * 32 + rr) * YROW + pc * 16);
      const long long mr = m0 + rb * 32 + rr;
      const unsigned off = mr < M ? (unsigned)((mr * N + n_even * SB + pc * 8) * 2) : OOB_OFFSET;
      __builtin_amdgcn_raw_buffer_store_b128(v, ydst, off, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

  load_b(0);
  load_a();
  store_b(0);

  // even slices accumulate in acc0, odd ones in acc1
  f32x16 acc0[RB], acc1[RB];
  // slice n into `acc`; `prev` (slice n-1, when FLUSH) leaves between its MFMAs.  FLUSH
  // and MORE are compile-time so that every wait is a counted one: with a branch around
  // the prefetch or the stores the compiler falls back to vmcnt(0) at the end of every
  // slice, which waits for the stores just issued and undoes the overlap.
  auto slice = [&](int n, f32x16 (&acc)[RB], const f32x16 (&prev)[RB], auto flush_c,
                   auto more_c, auto odd_c) {
    constexpr bool FLUSH = decltype(flush_c)::value, MORE = decltype(more_c)::value;
    constexpr bool ODD = decltype(odd_c)::value;   // n is odd: prev is an even slice
    __syncthreads();  // slice n is in its buffer; the other buffer is free again
    if constexpr (MORE) load_b(n + 1);
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[rb][e] = 0.f;
    const TA *brow = reinterpret_cast<const TA *>(s_b + (n & 1) * B_BYTES) + i * LDB + h * EPL;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const frag_t fb = *reinterpret_cast<const frag_t *>(brow + q * 2 * EPL);
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
        Mma<TA>::mma(fb, fa[rb][q], acc[rb]);   // rows = outputs, columns = pixels
      if constexpr (FLUSH) {
        if (q < 2 * RB) {
          if (q % 2 == 0)
            stage_write(prev[q / 2], q / 2, ODD ? 0 : 1);
          else if (!ODD)
            stage_store(n - 2, q / 2);
        }
      }
    }
    if constexpr (MORE) store_b((n + 1) & 1);
  };
  constexpr std::true_type yes{};
  constexpr std::false_type no{};
  // nslices is even (the host keeps slices_per_block even, N is a multiple of 64)
  slice(0, acc0, acc1, no, yes, no);
  int n = 1;
  for (; n + 2 <= nslices - 1; n += 2) {   // n and n + 1 are both followed by another slice
    slice(n, acc1, acc0, yes, yes, yes);
    slice(n + 1, acc0, acc1, yes, yes, no);
  }
  slice(n, acc1, acc0, yes, no, yes);       // n == nslices - 1
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    stage_write(acc1[rb], rb, 1);
    stage_store(nslices - 2, rb);
  }
}

// ---- Y = X * Wq^T for float32 tensors, on the bf16 matrix cores in three passes ------
// x [M][Cp] float32, wq = two bf16 planes [N][Cp] (high parts, then low parts), y [M][N]
// float32.  Every float is split as v = hi + lo with hi = bf16(v), lo = bf16(v - hi)
// (|v - hi - lo| <= 2^-18 |v|), and x*w is formed as hi*hi + hi*lo + lo*hi with float32
// accumulation: products exact, the dropped lo*lo term and the split residues are
// <= 3 * 2^-18 relative per product (measured 4e-6 of the output scale at K = 1792) -
// float32-class accuracy at 3/16 of the float32 MFMA's time (it runs at the vector rate).
// Non-finite inputs give NaN (inf - inf in the split) where exact arithmetic may give inf.
// Same structure as cts_gemm_kernel with one 32-row block per wavefront (its high and low
// fragments fill the registers two blocks took there) and 32-column slices.
template <int KH>
__global__ __launch_bounds__(GEMM_WAVES * 64, 2) void cts_gemm_split_kernel(
    const float *__restrict__ x, const bf16_t *__restrict__ wq, float *__restrict__ y,
    long long M, int Cp, int N, int slices_per_block, int ablate) {
  constexpr int EPL = 8;
  constexpr int NQ = KH / 16;
  constexpr int LDB = KH + EPL;
  constexpr int SB = 32;                      // output columns per slice
  constexpr int P_BYTES = SB * LDB * 2;       // one plane of one slice
  constexpr int B_BYTES = 2 * P_BYTES;        // high + low
  constexpr int NT = GEMM_WAVES * 64;
  constexpr int YROW = SB * 4 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char s_b[2 * B_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_y[GEMM_WAVES * 32 * YROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long m0 = ((long long)blockIdx.x * GEMM_WAVES + wid) * 32;
  const long long r = m0 + i;
  const float *xrow = x + (r < M ? r : M - 1) * Cp + h * EPL;

  const int kchunks = Cp / KH;
  const int slice0 = blockIdx.y * slices_per_block;
  const int nslices = min(slices_per_block, N / SB - slice0);
  const int total = nslices * kchunks;
  const bf16_t *wq_lo = wq + (size_t)N * Cp;
  y += (size_t)slice0 * SB;

  constexpr int PPP = SB * (KH / EPL);        // 16-byte pieces per plane and slice
  constexpr int BPT = 2 * PPP / NT;
  static_assert((2 * PPP) % NT == 0, "weight slice must split evenly over the threads");
  u32x4 breg[BPT];
  auto load_b = [&](int it) {
    const int n = slice0 + it / kchunks, kc = it % kchunks;
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NT;
      const int pl = q / PPP, q2 = q - pl * PPP;
      const int rr = q2 / (KH / EPL), pc = q2 - rr * (KH / EPL);
      breg[j] = ldg128((pl ? wq_lo : wq) + ((size_t)(n * SB + rr) * Cp + kc * KH + pc * EPL));
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NT;
      const int pl = q / PPP, q2 = q - pl * PPP;
      const int rr = q2 / (KH / EPL), pc = q2 - rr * (KH / EPL);
      *reinterpret_cast<u32x4 *>(s_b + buf * B_BYTES + pl * P_BYTES +
                                 (rr * LDB + pc * EPL) * 2) = breg[j];
    }
  };

  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x8 fa_hi[NQ], fa_lo[NQ];
  auto load_a = [&](int kc) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const u32x4 v0 = ldg128(xrow + kc * KH + q * 16);
      const u32x4 v1 = ldg128(xrow + kc * KH + q * 16 + 4);
      unsigned int hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = __uint_as_float(e < 2 ? v0[2 * e] : v1[2 * e - 4]);
        const float a1 = __uint_as_float(e < 2 ? v0[2 * e + 1] : v1[2 * e - 3]);
        const bf16x2 ph = {(__bf16)a0, (__bf16)a1};
        const unsigned int uh = __builtin_bit_cast(unsigned int, ph);
        const float r0 = a0 - __uint_as_float(uh << 16);
        const float r1 = a1 - __uint_as_float(uh & 0xffff0000u);
        const bf16x2 pl = {(__bf16)r0, (__bf16)r1};
        hi[e] = uh;
        lo[e] = __builtin_bit_cast(unsigned int, pl);
      }
      const u32x4 ph4 = {hi[0], hi[1], hi[2], hi[3]};
      const u32x4 pl4 = {lo[0], lo[1], lo[2], lo[3]};
      fa_hi[q] = __builtin_bit_cast(bf16x8, ph4);
      fa_lo[q] = __builtin_bit_cast(bf16x8, pl4);
    }
  };

  // The finished 32x32 block of slice n leaves while slice n+1 is contracted: two
  // accumulators alternate, and the staging writes / row reads / global stores of the
  // previous block sit between the MFMAs of the next one (both waves of a SIMD stop at the
  // same workgroup barriers, so work left between barrier and MFMAs idles the matrix pipe).
  unsigned char *ys = s_y + wid * 32 * YROW;
  const __amdgpu_buffer_rsrc_t ydst = __builtin_amdgcn_make_buffer_rsrc(
      y, 0, (int)(((size_t)M * N - (size_t)slice0 * SB) * 4), 0x00020000);
  // lane (i, h), registers 4g..4g+3 hold outputs 8g + 4h + 0..3 of pixel i
  auto stage_write = [&](const f32x16 &acc) {
    float *yr = reinterpret_cast<float *>(ys + i * YROW) + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
      *reinterpret_cast<f32x4 *>(yr + 8 * g) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto stage_store = [&](int n) {
    const int pc = lane & 7, rq = lane >> 3;   // 8 lanes x 16 B per 32-float row
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rr = ps * 8 + rq;
      const u32x4 v = *reinterpret_cast<const u32x4 *>(ys + rr * YROW + pc * 16);
      const long long mr = m0 + rr;
      // buffer stores without a branch (rows past M fall outside the window and are
      // dropped): the compiler can then count them, and the wait for the weight loads
      // issued before them does not wait for these stores to complete
      const unsigned off = mr < M ? (unsigned)((mr * N + n * SB + pc * 4) * 4) : OOB_OFFSET;
      __builtin_amdgcn_raw_buffer_store_b128(v, ydst, off, 0, 0);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };

#ifdef GANET_ABLATE
  unsigned long long ts[80];
  int nts = 0;
  const bool stamp = (ablate & 32) && blockIdx.x == gridDim.x / 2 && tid == 0;
#define GANET_STAMP() do { if (stamp && nts < 80) ts[nts++] = wall_clock64(); } while (0)
#else
#define GANET_STAMP() do {} while (0)
#endif
  GANET_STAMP();
  load_b(0);
  load_a(0);
  store_b(0);
#ifdef GANET_ABLATE
  if (ablate & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
  GANET_STAMP();

  f32x16 acc0, acc1;
  // slice n into `acc`; `prev` (slice n-1, when flush) leaves during the first K chunk
  auto slice = [&](int n, f32x16 &acc, const f32x16 &prev, bool flush) {
    for (int kc = 0; kc < kchunks; ++kc) {
      const int it = n * kchunks + kc;
      const bool more = it + 1 < total;
      __syncthreads();  // slice `it` is in its buffer; the other buffer is free again
      GANET_STAMP();
      if (more) load_b(it + 1);
      if (kchunks > 1 && it > 0) load_a(kc);
      if (kc == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      }
      const bool out_prev = flush && kc == 0;
      const unsigned char *Bs = s_b + (it & 1) * B_BYTES;
      const bf16_t *bh = reinterpret_cast<const bf16_t *>(Bs) + i * LDB + h * EPL;
      const bf16_t *bl = reinterpret_cast<const bf16_t *>(Bs + P_BYTES) + i * LDB + h * EPL;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bf16x8 fh = *reinterpret_cast<const bf16x8 *>(bh + q * 16);
        const bf16x8 fl = *reinterpret_cast<const bf16x8 *>(bl + q * 16);
#ifdef GANET_ABLATE  // profiling build only: bit0 skips the MFMAs, operands stay live
        if (ablate & 1) {
          asm volatile("" ::"v"(fa_hi[q]), "v"(fa_lo[q]), "v"(fh), "v"(fl));
          continue;
        }
#endif
        // rows = outputs, columns = pixels; small terms first
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, fa_hi[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fa_lo[q], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fa_hi[q], acc, 0, 0, 0);
        if (q == 0 && out_prev) stage_write(prev);
        if (q == (NQ > 1 ? 1 : 0) && out_prev) stage_store(n - 1);
      }
      GANET_STAMP();
      if (more) store_b((it + 1) & 1);
      GANET_STAMP();
    }
  };
#ifdef GANET_ABLATE
  const bool all_stores = !(ablate & 4);   // bit2: keep only the last slice's stores
#else
  constexpr bool all_stores = true;
#endif
  int n = 0;
  for (; n + 1 < nslices; n += 2) {
    slice(n, acc0, acc1, n > 0 && all_stores);
    slice(n + 1, acc1, acc0, all_stores);
  }
  if (n < nslices) {
    slice(n, acc0, acc1, n > 0 && all_stores);
    stage_write(acc0);
  } else {
    stage_write(acc1);
  }
  stage_store(nslices - 1);
  GANET_STAMP();
#ifdef GANET_ABLATE
  if (stamp) {
    printf("split gemm stamps (10 ns units, block %d):", (int)blockIdx.x);
    for (int j = 1; j < nts; ++j) printf(" %d", (int)(ts[j] - ts[0]));
    printf("\n");
  }
#endif
#undef GANET_STAMP
}

// ---- Y (+)= X * Wq^T for float32 tensors with a long K and few output columns ----------
// (the backward's grad_x = grad_Y * Wq: K = T*Cout = 448, N = Cin = 256).  Three bf16
// passes over high/low splits like cts_gemm_split_kernel, but with the loops the other way
// round: a wavefront owns 32 rows and keeps the accumulators of ALL NS 32-column slices
// live while it walks K in chunks of 64, so x is read once (the slice-outer kernel reads it
// once per slice - 8 times here, which is why it measured no faster than the float32 MFMA).
// The weight chunk of all slices (N x 64 x two planes) sits in LDS, one chunk at a time.
template <int NS>
__global__ __launch_bounds__(GEMM_WAVES * 64, 2) void cts_gemm_split_wide_kernel(
    const float *__restrict__ x, const bf16_t *__restrict__ wq, float *__restrict__ y,
    long long M, int K, int ldy, int accumulate) {
  constexpr int KC = 64, NQ = KC / 16, EPL = 8;
  constexpr int N = NS * 32;
  constexpr int LDB = KC + EPL;                 // +16 B: conflict-free ds_read_b128
  constexpr int P_BYTES = N * LDB * 2;          // one plane of one chunk
  constexpr int NT = GEMM_WAVES * 64;
  constexpr int YROW = 32 * 4 + 16;
  __shared__ __attribute__((aligned(16))) unsigned char s_b[2 * P_BYTES];
  __shared__ __attribute__((aligned(16))) unsigned char s_y[GEMM_WAVES * 32 * YROW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 31, h = lane >> 5;
  const long long m0 = ((long long)blockIdx.x * GEMM_WAVES + wid) * 32;
  const long long r = m0 + i;
  const float *xrow = x + (r < M ? r : M - 1) * K + h * EPL;
  const bf16_t *wq_lo = wq + (size_t)N * K;
  const int kchunks = K / KC;

  constexpr int PPP = N * (KC / EPL);           // 16-byte pieces per plane and chunk
  constexpr int BPT = 2 * PPP / NT;
  static_assert((2 * PPP) % NT == 0, "weight chunk must split evenly over the threads");
  auto stage_b = [&](int kc) {
#pragma unroll
    for (int j = 0; j < BPT; ++j) {
      const int q = tid + j * NT;
      const int pl = q / PPP, q2 = q - pl * PPP;
      const int rr = q2 / (KC / EPL), pc = q2 - rr * (KC / EPL);
      const u32x4 v = ldg128((pl ? wq_lo : wq) + ((size_t)rr * K + kc * KC + pc * EPL));
      *reinterpret_cast<u32x4 *>(s_b + pl * P_BYTES + (rr * LDB + pc * EPL) * 2) = v;
    }
  };
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  u32x4 raw[2 * NQ];
  auto load_a = [&](int kc) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      raw[2 * q] = ldg128(xrow + kc * KC + q * 16);
      raw[2 * q + 1] = ldg128(xrow + kc * KC + q * 16 + 4);
    }
  };
  bf16x8 fa_hi[NQ], fa_lo[NQ];
  auto split_a = [&]() {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      unsigned int hi[4], lo[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const u32x4 &v = e < 2 ? raw[2 * q] : raw[2 * q + 1];
        const float a0 = __uint_as_float(v[(2 * e) & 3]);
        const float a1 = __uint_as_float(v[(2 * e + 1) & 3]);
        const bf16x2 ph = {(__bf16)a0, (__bf16)a1};
        const unsigned int uh = __builtin_bit_cast(unsigned int, ph);
        const float r0 = a0 - __uint_as_float(uh << 16);
        const float r1 = a1 - __uint_as_float(uh & 0xffff0000u);
        const bf16x2 pl = {(__bf16)r0, (__bf16)r1};
        hi[e] = uh;
        lo[e] = __builtin_bit_cast(unsigned int, pl);
      }
      const u32x4 ph4 = {hi[0], hi[1], hi[2], hi[3]};
      const u32x4 pl4 = {lo[0], lo[1], lo[2], lo[3]};
      fa_hi[q] = __builtin_bit_cast(bf16x8, ph4);
      fa_lo[q] = __builtin_bit_cast(bf16x8, pl4);
    }
  };

  f32x16 acc[NS];
#pragma unroll
  for (int sl = 0; sl < NS; ++sl)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[sl][e] = 0.f;

  load_a(0);
  for (int kc = 0; kc < kchunks; ++kc) {
    __syncthreads();            // the previous chunk's weights have been read
    stage_b(kc);
    split_a();
    if (kc + 1 < kchunks) load_a(kc + 1);
    __syncthreads();
    const bf16_t *bh = reinterpret_cast<const bf16_t *>(s_b) + i * LDB + h * EPL;
    const bf16_t *bl = reinterpret_cast<const bf16_t *>(s_b + P_BYTES) + i * LDB + h * EPL;
#pragma unroll
    for (int sl = 0; sl < NS; ++sl)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bf16x8 fh = *reinterpret_cast<const bf16x8 *>(bh + sl * 32 * LDB + q * 16);
        const bf16x8 fl = *reinterpret_cast<const bf16x8 *>(bl + sl * 32 * LDB + q * 16);
        // rows = outputs, columns = pixels; small terms first
        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fl, fa_hi[q], acc[sl], 0, 0, 0);
        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fa_lo[q], acc[sl], 0, 0, 0);
        acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fh, fa_hi[q], acc[sl], 0, 0, 0);
      }
  }

  // lane (i, h), registers 4g..4g+3 of slice sl hold outputs sl*32 + 8g + 4h + 0..3 of pixel i
  unsigned char *ys = s_y + wid * 32 * YROW;
  const int pc = lane & 7, rq = lane >> 3;       // 8 lanes x 16 B per 32-float row
  // when accumulating, the destination's old values of slice sl + 1 are fetched while slice
  // sl is staged and stored (one exposed read latency per wavefront instead of NS)
  f32x4 old_v[4], next_v[4];
  auto fetch_old = [&](int sl, f32x4 (&o)[4]) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const long long mr = m0 + ps * 8 + rq;
      const u32x4 raw_o = ldg128(y + (mr < M ? mr : M - 1) * ldy + sl * 32 + pc * 4);
      o[ps] = __builtin_bit_cast(f32x4, raw_o);
    }
  };
  if (accumulate) fetch_old(0, old_v);
#pragma unroll
  for (int sl = 0; sl < NS; ++sl) {
    if (accumulate && sl + 1 < NS) fetch_old(sl + 1, next_v);
    float *yr = reinterpret_cast<float *>(ys + i * YROW) + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 v = {acc[sl][4 * g], acc[sl][4 * g + 1], acc[sl][4 * g + 2], acc[sl][4 * g + 3]};
      *reinterpret_cast<f32x4 *>(yr + 8 * g) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int rr = ps * 8 + rq;
      f32x4 v = *reinterpret_cast<const f32x4 *>(ys + rr * YROW + pc * 16);
      const long long mr = m0 + rr;
      if (mr < M) {
        float *dst = y + mr * ldy + sl * 32 + pc * 4;
        if (accumulate) v += old_v[ps];
        *reinterpret_cast<f32x4 *>(dst) = v;
      }
    }
    if (accumulate && sl + 1 < NS) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) old_v[ps] = next_v[ps];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// ---- out = sum over taps of bilinear samples of Y -----------------------------------
struct SampleArgs {
  const void *y;        // [nimg*H*W][T*CoutP], first image of the chunk at row 0
  const float *offset;  // [B][2T][Ho][Wo]  (whole batch)
  const float *mask;    // [B][T][Ho][Wo] or nullptr
  const void *bias;     // [Cout] or nullptr
  void *out;            // whole batch
  int img0, nimg;       // images of this chunk
  int H, W, Ho, Wo, Cout, CoutP, T;
  int kW, sH, sW, pH, pW, dH, dW;
  int out_nhwc, relu, points, ablate;
};

// one 16-byte piece of a corner row, weight w, into fp32 accumulators
template <typename TY>
__device__ __forceinline__ void acc_piece(const u32x4 &v, float w, f32x2 *acc);
template <>
__device__ __forceinline__ void acc_piece<float>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const f32x2 d = {__uint_as_float(v[2 * j]), __uint_as_float(v[2 * j + 1])};
    acc[j] = d * w + acc[j];
  }
}
template <>
__device__ __forceinline__ void acc_piece<bf16_t>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 d = {__uint_as_float(v[j] << 16), __uint_as_float(v[j] & 0xffff0000u)};
    acc[j] = d * w + acc[j];
  }
}
template <>
__device__ __forceinline__ void acc_piece<f16_t>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int u = v[j];
    const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
    const _Float16 hi = __builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    const f32x2 d = {(float)lo, (float)hi};
    acc[j] = d * w + acc[j];
  }
}

template <typename TY, typename TO>
__global__ __launch_bounds__(SAMPLE_WAVES * 64) void cts_sample_kernel(SampleArgs a) {
  constexpr int ESY = sizeof(TY);
  constexpr int EPL = 16 / ESY;            // channels per lane
  constexpr int LPX = 64 / EPL;            // lanes per pixel (64-channel block)
  constexpr int PPW = 64 / LPX;            // pixels per wavefront pass
  constexpr int NACC = EPL / 2;
  __shared__ __attribute__((aligned(16))) unsigned int s_off[SAMPLE_WAVES][PPW][TAP_BLOCK][4];
  __shared__ __attribute__((aligned(16))) float s_w[SAMPLE_WAVES][PPW][TAP_BLOCK][4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int pj = lane / LPX, piece = lane % LPX;
  const int cb = blockIdx.y;               // 64-channel block of the output
  const int HWo = a.Ho * a.Wo;
  const long long P = (long long)a.nimg * HWo;
  const int N = a.T * a.CoutP;
  const unsigned px_bytes = (unsigned)N * ESY;            // one pixel row of Y
  const unsigned row_bytes = (unsigned)a.W * px_bytes;
  const unsigned img_bytes = (unsigned)a.H * row_bytes;
  const size_t y_bytes = (size_t)a.nimg * img_bytes;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void *>(a.y), 0, (int)(y_bytes < 0x7fffffffull ? y_bytes : 0x7fffffffull),
      0x00020000);
  const unsigned lane_off = (unsigned)((cb * 64 + piece * EPL) * ESY);
  const TO *bias = static_cast<const TO *>(a.bias);
  TO *out = static_cast<TO *>(a.out);

  // Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8).  Every XCD walks one
  // contiguous eighth of the pixels, so the rows of Y shared by neighbouring pixels
  // (the next column, the next image row) are fetched into ONE L2 instead of several.
  const long long ngroups = (P + PPW - 1) / PPW;
  const long long per_xcd = (ngroups + 7) / 8;
  const int xcd = blockIdx.x & 7;
  const long long xbeg = xcd * per_xcd;
  const long long xend = xbeg + per_xcd < ngroups ? xbeg + per_xcd : ngroups;
  const long long xstride = (long long)(gridDim.x >> 3) * SAMPLE_WAVES;
  // offsets (and mask) of this lane's taps of one pixel group, tap block t0
  constexpr int KPL = TAP_BLOCK / LPX;     // taps per lane and block
  auto load_offsets = [&](long long grp, int t0, float *oh, float *ow, float *mk) {
    const long long p = grp * PPW + pj;
    const bool valid = p < P && grp < xend;
    const int bl = valid ? (int)(p / HWo) : 0;
    const int hw = valid ? (int)(p - (long long)bl * HWo) : 0;
    const int b = a.img0 + bl;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const int t = t0 + piece + k * LPX;
      oh[k] = ow[k] = 0.f;
      mk[k] = 1.f;
      if (valid && t < a.T) {
        const float *op = a.offset + ((size_t)b * 2 * a.T + 2 * t) * HWo + hw;
        oh[k] = ldg_f32(op);
        ow[k] = ldg_f32(op + HWo);
        if (a.mask) mk[k] = ldg_f32(a.mask + ((size_t)b * a.T + t) * HWo + hw);
      }
    }
  };
  // the first tap block's offsets of the NEXT group are fetched while this group's
  // gathers are in flight: a wave walks only a handful of groups, each a chain of
  // offsets -> coordinates -> gather -> store, and this takes one link off it
  float nh[KPL], nw[KPL], nm[KPL];
  const long long grp0 = xbeg + (long long)(blockIdx.x >> 3) * SAMPLE_WAVES + wid;
  load_offsets(grp0, 0, nh, nw, nm);
  for (long long grp = grp0; grp < xend; grp += xstride) {
    const long long p = grp * PPW + pj;
    const bool valid = p < P;
    const int bl = valid ? (int)(p / HWo) : 0;            // image within the chunk
    const int hw = valid ? (int)(p - (long long)bl * HWo) : 0;
    const int ho = hw / a.Wo, wo = hw - ho * a.Wo;
    const int b = a.img0 + bl;

    f32x2 acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x2{0.f, 0.f};

    for (int t0 = 0; t0 < a.T; t0 += TAP_BLOCK) {
      const int nt = min(TAP_BLOCK, a.T - t0);
      // the LPX lanes of a pixel split the taps of this block
      float ch[KPL], cw[KPL], cm[KPL];
      if (t0 == 0) {
#pragma unroll
        for (int k = 0; k < KPL; ++k) {
          ch[k] = nh[k];
          cw[k] = nw[k];
          cm[k] = nm[k];
        }
      } else {
        load_offsets(grp, t0, ch, cw, cm);
      }
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const int tt = piece + k * LPX;
        if (tt >= nt) break;
        const int t = t0 + tt;
        const int ti = t / a.kW, tj = t - ti * a.kW;
        unsigned c1 = OOB_OFFSET, c2 = OOB_OFFSET, c3 = OOB_OFFSET, c4 = OOB_OFFSET;
        float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
        if (valid) {
          const float off_h = ch[k];
          float off_w = cw[k];
          const float mk = cm[k];
          if (a.points)  // dcn_offset = pts - base  (lanepoints_conv.py:63-74,170)
            off_w = off_w - (float)(tj - a.pW);
          const float h_im = (float)(ho * a.sH - a.pH + ti * a.dH) + off_h;
          const float w_im = (float)(wo * a.sW - a.pW + tj * a.dW) + off_w;
          if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
            const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
            const int h_high = h_low + 1, w_high = w_low + 1;
            const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
            const float hh = 1.f - lh, hw_ = 1.f - lw;
            w1 = hh * hw_ * mk;
            w2 = hh * lw * mk;
            w3 = lh * hw_ * mk;
            w4 = lh * lw * mk;
            const unsigned base = (unsigned)bl * img_bytes + (unsigned)h_low * row_bytes +
                                  (unsigned)w_low * px_bytes +
                                  (unsigned)t * (unsigned)(a.CoutP * ESY);
            const bool hl = h_low >= 0, hh_ok = h_high <= a.H - 1;
            const bool wl = w_low >= 0, wh_ok = w_high <= a.W - 1;
            if (hl && wl) c1 = base;
            if (hl && wh_ok) c2 = base + px_bytes;
            if (hh_ok && wl) c3 = base + row_bytes;
            if (hh_ok && wh_ok) c4 = base + row_bytes + px_bytes;
          }
        }
        *reinterpret_cast<uint4 *>(s_off[wid][pj][tt]) = make_uint4(c1, c2, c3, c4);
        *reinterpret_cast<float4 *>(s_w[wid][pj][tt]) = make_float4(w1, w2, w3, w4);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (t0 == 0) load_offsets(grp + xstride, 0, nh, nw, nm);
#pragma unroll 2
      for (int tt = 0; tt < nt; ++tt) {
        const uint4 cx = *reinterpret_cast<const uint4 *>(s_off[wid][pj][tt]);
        const float4 w = *reinterpret_cast<const float4 *>(s_w[wid][pj][tt]);
#ifdef GANET_ABLATE  // bit3: no gather (weights and coordinates still formed)
        if (a.ablate & 8) {
          acc[0] = acc[0] + f32x2{w.x + w.y, w.z + w.w} * __uint_as_float(cx.x ^ cx.y ^ cx.z ^ cx.w);
          continue;
        }
#endif
        const u32x4 v1 = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.x + lane_off, 0, 0);
        const u32x4 v2 = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.y + lane_off, 0, 0);
        const u32x4 v3 = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.z + lane_off, 0, 0);
        const u32x4 v4 = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.w + lane_off, 0, 0);
        acc_piece<TY>(v1, w.x, acc);
        acc_piece<TY>(v2, w.y, acc);
        acc_piece<TY>(v3, w.z, acc);
        acc_piece<TY>(v4, w.w, acc);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    if (valid) {
      const int c0 = cb * 64 + piece * EPL;
      float r[EPL];
#pragma unroll
      for (int j = 0; j < NACC; ++j) {
        r[2 * j] = acc[j].x;
        r[2 * j + 1] = acc[j].y;
      }
#pragma unroll
      for (int e = 0; e < EPL; ++e) {
        if (bias && c0 + e < a.Cout) r[e] += to_f32(bias[c0 + e]);
        if (a.relu) r[e] = fmaxf(r[e], 0.f);
      }
      const long long pg = (long long)b * HWo + hw;   // pixel index in the whole batch
      if (a.out_nhwc) {
        TO *dst = out + pg * a.Cout + c0;
        if (c0 + EPL <= a.Cout && (a.Cout * sizeof(TO)) % 16 == 0 && sizeof(TO) == ESY) {
          u32x4 pk;
          TO tmp[EPL];
#pragma unroll
          for (int e = 0; e < EPL; ++e) tmp[e] = from_f32<TO>(r[e]);
          __builtin_memcpy(&pk, tmp, 16);
          *reinterpret_cast<u32x4 *>(dst) = pk;
        } else {
#pragma unroll
          for (int e = 0; e < EPL; ++e)
            if (c0 + e < a.Cout) dst[e] = from_f32<TO>(r[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
          if (c0 + e < a.Cout)
            out[((size_t)b * a.Cout + c0 + e) * HWo + hw] = from_f32<TO>(r[e]);
      }
    }
  }
}

template <typename TA, typename TY, int RB>
void launch_gemm_rb(int kh, const void *x, const void *wq, void *y, long long M, int Cp,
                    int N, int ablate, hipStream_t s) {
  constexpr int KBIG = sizeof(TA) == 4 ? 128 : 256;
  constexpr int ROWS = GEMM_WAVES * 32 * RB;
  const unsigned gx = (unsigned)((M + ROWS - 1) / ROWS);
  const int nslices = N / 64;
  int ny = gx > 64 ? 1 : (int)((256 + gx - 1) / gx);
  if (ny > nslices) ny = nslices;
  const int per = (nslices + ny - 1) / ny;
  ny = (nslices + per - 1) / per;
  dim3 grid(gx, (unsigned)ny);
  dim3 block(GEMM_WAVES * 64);
  const TA *xx = static_cast<const TA *>(x);
  const TA *ww = static_cast<const TA *>(wq);
  TY *yy = static_cast<TY *>(y);
  if (kh == KBIG)
    hipLaunchKernelGGL((cts_gemm_kernel<TA, TY, KBIG, RB>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per, ablate);
  else if (kh == KBIG / 2)
    hipLaunchKernelGGL((cts_gemm_kernel<TA, TY, KBIG / 2, RB>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per, ablate);
  else
    hipLaunchKernelGGL((cts_gemm_kernel<TA, TY, KBIG / 4, RB>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per, ablate);
}

template <typename TA, typename TY>
void launch_gemm(int kh, const void *x, const void *wq, void *y, long long M, int Cp,
                 int N, int ablate, hipStream_t s) {
  // two row blocks per wavefront once that still leaves a workgroup for most CUs
  if (M >= 192ll * GEMM_WAVES * 64)
    launch_gemm_rb<TA, TY, 2>(kh, x, wq, y, M, Cp, N, ablate, s);
  else
    launch_gemm_rb<TA, TY, 1>(kh, x, wq, y, M, Cp, N, ablate, s);
}

// 16-bit operands: the overlapped-epilogue kernel (GANET_CTS_GEMM=tile selects the
// tile-per-slice kernel above, for A/B runs)
template <typename TA, typename TY, int RB>
void launch_gemm_ov_rb(int kh, const void *x, const void *wq, void *y, long long M, int Cp,
                       int N, hipStream_t s) {
  constexpr int ROWS = GEMM_WAVES * 32 * RB;
  const unsigned gx = (unsigned)((M + ROWS - 1) / ROWS);
  const int nslices = N / 32;
  int ny = gx > 64 ? 1 : (int)((256 + gx - 1) / gx);
  if (ny > nslices / 2) ny = nslices / 2;
  int per = (nslices + ny - 1) / ny;
  per += per & 1;   // slices leave in pairs (whole 128-byte lines)
  ny = (nslices + per - 1) / per;
  dim3 grid(gx, (unsigned)ny);
  dim3 block(GEMM_WAVES * 64);
  const TA *xx = static_cast<const TA *>(x);
  const TA *ww = static_cast<const TA *>(wq);
  TY *yy = static_cast<TY *>(y);
  if (kh == 256)
    hipLaunchKernelGGL((cts_gemm_ov_kernel<TA, TY, 256, RB>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per);
  else if (kh == 128)
    hipLaunchKernelGGL((cts_gemm_ov_kernel<TA, TY, 128, RB>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per);
  else
    hipLaunchKernelGGL((cts_gemm_ov_kernel<TA, TY, 64, RB>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per);
}

bool gemm_overlapped() {
  static const int v = [] {
    const char *e = getenv("GANET_CTS_GEMM");
    return (e && e[0] == 't') ? 0 : 1;
  }();
  return v != 0;
}

template <typename TA, typename TY>
void launch_gemm16(int kh, const void *x, const void *wq, void *y, long long M, int Cp,
                   int N, int ablate, hipStream_t s) {
  if (!gemm_overlapped() || ablate || Cp != kh)
    launch_gemm<TA, TY>(kh, x, wq, y, M, Cp, N, ablate, s);
  else if (M >= 192ll * GEMM_WAVES * 64)
    launch_gemm_ov_rb<TA, TY, 2>(kh, x, wq, y, M, Cp, N, s);
  else
    launch_gemm_ov_rb<TA, TY, 1>(kh, x, wq, y, M, Cp, N, s);
}

void launch_gemm_split(const void *x, const void *wq, void *y, long long M, int Cp, int N,
                       int ablate, hipStream_t s) {
  constexpr int ROWS = GEMM_WAVES * 32;
  const unsigned gx = (unsigned)((M + ROWS - 1) / ROWS);
  const int nslices = N / 32;
  int ny = gx > 64 ? 1 : (int)((256 + gx - 1) / gx);
  if (ny > nslices) ny = nslices;
  const int per = (nslices + ny - 1) / ny;
  ny = (nslices + per - 1) / per;
  dim3 grid(gx, (unsigned)ny);
  dim3 block(GEMM_WAVES * 64);
  const float *xx = static_cast<const float *>(x);
  const bf16_t *ww = static_cast<const bf16_t *>(wq);
  float *yy = static_cast<float *>(y);
  if (Cp % 256 == 0)
    hipLaunchKernelGGL((cts_gemm_split_kernel<256>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per, ablate);
  else if (Cp % 128 == 0)
    hipLaunchKernelGGL((cts_gemm_split_kernel<128>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per, ablate);
  else
    hipLaunchKernelGGL((cts_gemm_split_kernel<64>), grid, block, 0, s, xx, ww, yy, M, Cp, N, per, ablate);
}

template <typename TY, typename TO>
void launch_sample(const SampleArgs &a, hipStream_t s) {
  constexpr int PPW = 64 / (64 / (16 / (int)sizeof(TY)));
  const long long P = (long long)a.nimg * a.Ho * a.Wo;
  const long long groups = (P + PPW - 1) / PPW;
  long long blocks = (groups + SAMPLE_WAVES - 1) / SAMPLE_WAVES;
  // a few pixel groups per wavefront amortise the launch; keep every CU busy
  const long long cap = 256ll * 16;
  if (blocks > cap) blocks = cap;
  blocks = (blocks + 7) / 8 * 8;   // the XCD walk needs a whole number of rounds
  dim3 grid((unsigned)blocks, (unsigned)(a.CoutP / 64));
  hipLaunchKernelGGL((cts_sample_kernel<TY, TO>), grid, dim3(SAMPLE_WAVES * 64), 0, s, a);
}

}  // namespace

// y[M][N] = x[M][K] * wq[N][K]^T in exact float32 (the backward's GEMMs: per-tap products,
// grad_x = grad_Y * Wq).  N % 64 == 0, K % 32 == 0.
int cts_gemm_f32(const float *x, const float *wq, float *y, long long M, int K, int N,
                 hipStream_t s) {
  if (M <= 0 || N % 64 || K % 32) return GANET_ERR_UNSUPPORTED;
  const int kh = K % 128 == 0 ? 128 : (K % 64 == 0 ? 64 : 32);
  launch_gemm<float, float>(kh, x, wq, y, M, K, N, 0, s);
  return ganet_check_launch();
}

// the same product as three bf16 passes (cts_gemm_split_kernel); wq_split = a high and a low
// bf16 plane [N][K] each.  N % 32 == 0, K % 64 == 0.
int cts_gemm_f32_split(const float *x, const void *wq_split, float *y, long long M, int K,
                       int N, hipStream_t s) {
  if (M <= 0 || N % 32 || K % 64) return GANET_ERR_UNSUPPORTED;
  launch_gemm_split(x, wq_split, y, M, K, N, 0, s);
  return ganet_check_launch();
}

// y[M][ldy] (+)= x[M][K] * wq^T, wq_split = two bf16 planes [N][K]; N in {64, 128, 192, 256},
// K % 64 == 0, ldy % 4 == 0: the K-outer kernel (x read once)
int cts_gemm_f32_split_wide(const float *x, const void *wq_split, float *y, long long M, int K,
                            int N, int ldy, int accumulate, hipStream_t s) {
  if (M <= 0 || K % 64 || N % 64 || N > 256 || ldy % 4) return GANET_ERR_UNSUPPORTED;
  const unsigned gx = (unsigned)((M + GEMM_WAVES * 32 - 1) / (GEMM_WAVES * 32));
  const bf16_t *ww = static_cast<const bf16_t *>(wq_split);
  const dim3 block(GEMM_WAVES * 64);
  switch (N / 32) {
    case 2: hipLaunchKernelGGL((cts_gemm_split_wide_kernel<2>), dim3(gx), block, 0, s, x, ww, y, M, K, ldy, accumulate); break;
    case 4: hipLaunchKernelGGL((cts_gemm_split_wide_kernel<4>), dim3(gx), block, 0, s, x, ww, y, M, K, ldy, accumulate); break;
    case 6: hipLaunchKernelGGL((cts_gemm_split_wide_kernel<6>), dim3(gx), block, 0, s, x, ww, y, M, K, ldy, accumulate); break;
    default: hipLaunchKernelGGL((cts_gemm_split_wide_kernel<8>), dim3(gx), block, 0, s, x, ww, y, M, K, ldy, accumulate); break;
  }
  return ganet_check_launch();
}

// Applicable when the commuted order is cheaper and its assumptions hold: one
// deformable group (one Y per tap) and fewer output than input channels.
bool cts_split(const ganet_dcn_params *p, int Cp) {
  return p->dtype == GANET_F32 && p->mfma_bf16 == 0 && Cp % 64 == 0;
}

bool cts_applicable(const ganet_dcn_params *p, int Cp, int CoutP) {
  // float32: with the float32 MFMA (it runs at the vector rate) the per-tap GEMM alone
  // costs what the fused kernel costs in total, so float32 tensors commute only when
  // the GEMM may run as three bf16 passes (cts_split) or the caller asks for it.
  if (p->dtype == GANET_F32 && p->mfma_bf16 != 1 && !cts_split(p, Cp) && p->algo != 2)
    return false;
  // one image's Y must stay below 2 GiB (32-bit byte offsets in the sampling kernel's
  // buffer window); larger maps take the fused kernel
  const size_t es = (p->dtype == GANET_F32 && p->mfma_bf16 != 1) ? 4 : 2;
  const size_t img_y = (size_t)p->H * p->W * p->kH * p->kW * CoutP * es;
  if (img_y >= 0x7fffffffull) return false;
  return p->deformable_groups == 1 && p->groups == 1 && 2 * CoutP <= Cp &&
         p->sH == 1 && p->sW == 1 && p->kH * p->kW <= 64;
}

// images per chunk: Y of one chunk stays well inside the 256 MiB Infinity Cache, the chunks
// are balanced so that no launch is left with a sliver of the batch, and their number is
// chosen for whole rounds of GEMM tiles over the 256 CUs: a chunk of 282 tiles costs a second
// round for 26 of them (C4 at 9 images per chunk: 1.62 ms against 1.40 at 8 = 250 tiles,
// profiles/r02_algo3_notes.txt).  `split`: float32 tensors on the three-pass GEMM.
int cts_chunk_images(const ganet_dcn_params *p, int CoutP, int ta, bool split) {
  const size_t es = ta == GANET_F32 ? 4 : 2;
  const size_t hw = (size_t)p->H * p->W;
  const size_t per_img = hw * p->kH * p->kW * CoutP * es;
  // (a knob that sizes the workspace is read ONCE per process: a query and the call it sizes
  // can then never see two values)
  static const size_t budget = [] {
    const char *e = getenv("GANET_CTS_BUDGET_MB");
    const long long mb = e ? atoll(e) : 0;
    return mb > 0 ? (size_t)mb << 20 : (size_t)144 << 20;
  }();
  size_t nmax = budget / (per_img ? per_img : 1);
  if (nmax < 1) nmax = 1;
  if (nmax > (size_t)p->B) nmax = p->B;
  const size_t B = (size_t)p->B;
  const size_t cmin = (B + nmax - 1) / nmax;
  // rounds of one chunk of n images: tiles of 512 pixels (two row blocks per wavefront) once
  // that fills the chip, 256 otherwise and on the three-pass kernel (launch_gemm*)
  auto rounds = [&](size_t n) {
    const size_t m = n * hw;
    const size_t rows = (!split && m >= 192ull * GEMM_WAVES * 64) ? GEMM_WAVES * 64 : GEMM_WAVES * 32;
    const size_t tiles = (m + rows - 1) / rows;
    return (tiles + 255) / 256;
  };
  size_t best_nb = (B + cmin - 1) / cmin;
  double best_cost = -1.0;
  const size_t cmax = 2 * cmin + 2 < B ? 2 * cmin + 2 : B;
  for (size_t c = cmin; c <= cmax; ++c) {
    const size_t nb = (B + c - 1) / c;
    const size_t full = B / nb, rest = B - full * nb;
    // a launch pair costs about a quarter of a round on top of its rounds
    const double cost = (double)(full * rounds(nb) + (rest ? rounds(rest) : 0)) +
                        0.25 * (double)(full + (rest ? 1 : 0));
    if (best_cost < 0 || cost < best_cost - 1e-9) {
      best_cost = cost;
      best_nb = nb;
    }
  }
  return (int)best_nb;
}

size_t cts_y_bytes(const ganet_dcn_params *p, int CoutP, int ta, int nb) {
  const size_t es = ta == GANET_F32 ? 4 : 2;
  const size_t per_img = (size_t)p->H * p->W * p->kH * p->kW * CoutP * es;
  return ganet_align_up(per_img * (size_t)nb, 256);
}

// xh: NHWC map in arithmetic dtype `ta` with Cp channels; wq: packed weights
// [T][CoutP][Cp] (same layout the fused kernel uses); ybuf: cts_y_bytes() of scratch.
int cts_forward(const void *xh, const float *offset, const float *mask, const void *wq,
                const void *bias, void *out, const ganet_dcn_params *p, int ta, int kc,
                int Cp, int CoutP, int Ho, int Wo, void *ybuf, int nb, hipStream_t s) {
  const bool split = ta == GANET_F32 && cts_split(p, Cp);
  const int T = p->kH * p->kW;
  const int N = T * CoutP;
  const size_t es = ta == GANET_F32 ? 4 : 2;
  const size_t img_y = (size_t)p->H * p->W * N * es;
  if (img_y >= 0x7fffffffull) return GANET_ERR_UNSUPPORTED;
  // nb images per chunk (cts_chunk_images, or the whole batch when Y is kept for the backward)
  if (nb < 1 || (size_t)nb * img_y >= 0x7fffffffull) return GANET_ERR_UNSUPPORTED;
  for (int img0 = 0; img0 < p->B; img0 += nb) {
    const int nimg = (p->B - img0) < nb ? (p->B - img0) : nb;
    const long long M = (long long)nimg * p->H * p->W;
    const unsigned char *xc =
        static_cast<const unsigned char *>(xh) + (size_t)img0 * p->H * p->W * Cp * es;
    if (split)
      launch_gemm_split(xc, wq, ybuf, M, Cp, N, p->debug, s);
    else if (ta == GANET_F32)
      launch_gemm<float, float>(kc, xc, wq, ybuf, M, Cp, N, p->debug, s);
    else if (ta == GANET_BF16)
      launch_gemm16<bf16_t, bf16_t>(kc, xc, wq, ybuf, M, Cp, N, p->debug, s);
    else
      launch_gemm16<f16_t, f16_t>(kc, xc, wq, ybuf, M, Cp, N, p->debug, s);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;

    SampleArgs a;
    a.y = ybuf;
    a.offset = offset;
    a.mask = mask;
    a.bias = bias;
    a.out = out;
    a.img0 = img0;
    a.nimg = nimg;
    a.H = p->H;
    a.W = p->W;
    a.Ho = Ho;
    a.Wo = Wo;
    a.Cout = p->Cout;
    a.CoutP = CoutP;
    a.T = T;
    a.kW = p->kW;
    a.sH = p->sH;
    a.sW = p->sW;
    a.pH = p->pH;
    a.pW = p->pW;
    a.dH = p->dH;
    a.dW = p->dW;
    a.out_nhwc = p->out_layout == GANET_NHWC;
    a.relu = p->fuse_relu;
    a.points = p->offset_is_points;
    a.ablate = p->debug;
    if (ta == GANET_F32)
      launch_sample<float, float>(a, s);
    else if (ta == GANET_BF16 && p->dtype == GANET_F32)
      launch_sample<bf16_t, float>(a, s);
    else if (ta == GANET_BF16)
      launch_sample<bf16_t, bf16_t>(a, s);
    else
      launch_sample<f16_t, f16_t>(a, s);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }
  return GANET_OK;
}

}  // namespace ganet_dcn
