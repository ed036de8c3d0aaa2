// Host-only: the size / plan queries of the library under UBSan + ASan with random and extreme
// parameters (tools/asan_plan.sh).  No launch happens in any of these calls; what is checked is
// that no signed overflow, bad shift or out-of-range conversion sits in the host arithmetic.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

#include "ganet_hip.h"

static std::mt19937_64 rng(99);
static int32_t pick(int kind) {
  static const int32_t extreme[] = {0, 1, -1, 2, 3, 7, 31, 32, 33, 63, 64, 65, 127, 128, 255, 256, 257, 1023,
                                    4096, 65535, 65536, 1 << 20, (1 << 24) + 1, 0x3fffffff, 0x7fffffff,
                                    (int32_t)0x80000000, -65536};
  switch (kind) {
    case 0: return (int32_t)(rng() % 9) + 1;
    case 1: return (int32_t)(rng() % 300) + 1;
    case 2: return extreme[rng() % (sizeof extreme / sizeof extreme[0])];
    default: return (int32_t)rng();
  }
}

int main() {
  unsigned long long acc = 0;
  for (int it = 0; it < 400000; ++it) {
    ganet_dcn_params p;
    std::memset(&p, 0, sizeof p);
    int32_t *f = reinterpret_cast<int32_t *>(&p);
    const int wild = it % 4;
    for (size_t i = 0; i < sizeof p / 4; ++i)
      f[i] = pick(wild == 0 ? (int)(rng() % 2) : (wild == 1 ? (int)(rng() % 3) : (int)(rng() % 4)));
    if (wild < 3) {   // mostly plausible enums so that the plans get past the first checks
      p.dtype = (int32_t)(rng() % 3);
      p.x_layout = (int32_t)(rng() % 2);
      p.out_layout = (int32_t)(rng() % 2);
      p.mfma_bf16 = (int32_t)(rng() % 3);
      p.fuse_relu = p.offset_is_points = p.weight_is_packed = p.debug = 0;
      static const int32_t algos[] = {0, 1, 2, 3, 4, 5, 6, 0x100, 0x102, 0x105, 0x200, 0x300, 0x400};
      p.algo = algos[rng() % 13];
      p.groups = (int32_t)(rng() % 3) + 1;
      p.deformable_groups = (int32_t)(rng() % 3) + 1;
    }
    acc += ganet_deform_conv_workspace_bytes(&p);
    acc += (unsigned)ganet_deform_conv_algorithm(&p);
    acc += ganet_deform_conv_status_offset(&p);
    acc += ganet_deform_conv_y_offset(&p);
    acc += ganet_deform_conv_packed_weight_bytes(&p);
    acc += ganet_deform_conv_backward_workspace_bytes(&p);
    acc += (unsigned)ganet_dcn_out_size(f[2], f[9], f[11], f[5], f[7]);

    ganet_decode_params d;
    std::memset(&d, 0, sizeof d);
    d.B = pick((int)(rng() % 4));
    d.H = pick((int)(rng() % 4));
    d.W = pick((int)(rng() % 4));
    d.max_kpts = pick((int)(rng() % 4));
    d.max_centers = pick((int)(rng() % 4));
    d.max_groups = pick((int)(rng() % 4));
    d.flags = (int32_t)(rng() % 32);
    acc += ganet_decode_workspace_bytes(&d);
  }
  std::printf("plan fuzz: 400000 parameter sets, checksum %llu\n", acc);
  return 0;
}
