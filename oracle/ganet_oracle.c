/*
 * TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C) of the reference's
 * algorithm for the GANet hot path.  It is the checker for the HIP kernels
 * and the "port" CPU baseline of bench.py.  Nothing under ganet_amd/ links,
 * loads or calls this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * Parity pinning:
 *   - decode + grouping: pinned bit-exactly against the reference's own Python
 *     (imported in the build container by oracle/ref_import.py) through the
 *     golden vectors in tests/golden/ (oracle/make_golden.py wrote them).
 *   - deform-conv forward: PARITY UNPINNED BY THE REFERENCE.  The reference's
 *     op is CUDA-only (mmdet/ops/dcn/src/deform_conv_ext.cpp:51-67 raises on
 *     CPU), its prebuilt .so is absent and it holds no golden vectors.  This
 *     restatement follows the CUDA source term for term and is cross-checked
 *     in tests/ against an independent PyTorch formulation and known-answer
 *     cases.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 * -ffp-contract=off matters: the decode arithmetic must round exactly like
 * numpy / CPython do.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Deformable convolution forward (DCNv1) and modulated forward (DCNv2)
 * ---------------------------------------------------------------------- */

/* mmdet/ops/dcn/src/cuda/deform_conv_cuda_kernel.cu:84-115
 * (deformable_im2col_bilinear) - same operations in the same order. */
static float oracle_bilinear(const float *bottom_data, int data_width,
                             int height, int width, float h, float w) {
  int h_low = (int)floorf(h);
  int w_low = (int)floorf(w);
  int h_high = h_low + 1;
  int w_high = w_low + 1;
  float lh = h - (float)h_low;
  float lw = w - (float)w_low;
  float hh = 1.0f - lh, hw = 1.0f - lw;
  float v1 = 0.f, v2 = 0.f, v3 = 0.f, v4 = 0.f;
  if (h_low >= 0 && w_low >= 0) v1 = bottom_data[h_low * data_width + w_low];
  if (h_low >= 0 && w_high <= width - 1)
    v2 = bottom_data[h_low * data_width + w_high];
  if (h_high <= height - 1 && w_low >= 0)
    v3 = bottom_data[h_high * data_width + w_low];
  if (h_high <= height - 1 && w_high <= width - 1)
    v4 = bottom_data[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* Output size rule: mmdet/ops/dcn/deform_conv.py:98-111 and
 * mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:108-111. */
int oracle_dcn_out_size(int in, int pad, int dil, int k, int stride) {
  return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
}

/* im2col of ONE image following deformable_im2col_gpu_kernel
 * (deform_conv_cuda_kernel.cu:199-241) / the modulated variant (:570-633).
 * col layout: [Cin*kH*kW][Hout*Wout].  mask may be NULL (DCNv1). */
static void oracle_im2col_one(const float *x, const float *offset,
                              const float *mask, int Cin, int H, int W, int kH,
                              int kW, int pH, int pW, int sH, int sW, int dH,
                              int dW, int dgroups, int Hout, int Wout,
                              float *col) {
  const int cpg = Cin / dgroups;
  const int HWo = Hout * Wout;
  for (int c = 0; c < Cin; ++c) {
    const int dg = c / cpg;
    const float *im = x + (size_t)c * H * W;
    const float *off = offset + (size_t)dg * 2 * kH * kW * HWo;
    const float *msk = mask ? mask + (size_t)dg * kH * kW * HWo : NULL;
    for (int h_col = 0; h_col < Hout; ++h_col) {
      for (int w_col = 0; w_col < Wout; ++w_col) {
        const int h_in = h_col * sH - pH;
        const int w_in = w_col * sW - pW;
        for (int i = 0; i < kH; ++i) {
          for (int j = 0; j < kW; ++j) {
            const int t = i * kW + j;
            const float offset_h =
                off[((size_t)(2 * t) * Hout + h_col) * Wout + w_col];
            const float offset_w =
                off[((size_t)(2 * t + 1) * Hout + h_col) * Wout + w_col];
            float val = 0.f;
            const float h_im = (float)(h_in + i * dH) + offset_h;
            const float w_im = (float)(w_in + j * dW) + offset_w;
            if (h_im > -1 && w_im > -1 && h_im < H && w_im < W)
              val = oracle_bilinear(im, W, H, W, h_im, w_im);
            if (msk)
              val = val * msk[((size_t)t * Hout + h_col) * Wout + w_col];
            col[((size_t)c * kH * kW + t) * HWo + h_col * Wout + w_col] = val;
          }
        }
      }
    }
  }
}

/* GEMM of one image: out[o][p] = sum_kk W[o][kk] * col[kk][p], per conv group
 * (deform_conv_cuda.cpp:227-236).  Accumulates in double when acc64 != 0
 * (the better reference); in float, k-ordered, otherwise. */
static void oracle_gemm_one(const float *weight, const float *col,
                            const float *bias, int Cin, int Cout, int kHW,
                            int groups, int HWo, int acc64, float *out) {
  const int Cig = Cin / groups, Cog = Cout / groups;
  const int Kg = Cig * kHW;
  double *acc = (double *)malloc(sizeof(double) * (size_t)HWo);
  float *accf = (float *)malloc(sizeof(float) * (size_t)HWo);
  for (int o = 0; o < Cout; ++o) {
    const int g = o / Cog;
    const float *wrow = weight + (size_t)o * Kg;
    const float *cg = col + (size_t)g * Kg * HWo;
    if (acc64) {
      for (int p = 0; p < HWo; ++p) acc[p] = bias ? (double)bias[o] : 0.0;
      for (int kk = 0; kk < Kg; ++kk) {
        const double wv = (double)wrow[kk];
        const float *cr = cg + (size_t)kk * HWo;
        for (int p = 0; p < HWo; ++p) acc[p] += wv * (double)cr[p];
      }
      for (int p = 0; p < HWo; ++p) out[(size_t)o * HWo + p] = (float)acc[p];
    } else {
      for (int p = 0; p < HWo; ++p) accf[p] = bias ? bias[o] : 0.f;
      for (int kk = 0; kk < Kg; ++kk) {
        const float wv = wrow[kk];
        const float *cr = cg + (size_t)kk * HWo;
        for (int p = 0; p < HWo; ++p) accf[p] += wv * cr[p];
      }
      for (int p = 0; p < HWo; ++p) out[(size_t)o * HWo + p] = accf[p];
    }
  }
  free(acc);
  free(accf);
}

/* deform_conv_forward_cuda (deform_conv_cuda.cpp:152-260) on NCHW float32.
 * mask/bias NULL -> DCNv1; non-NULL mask -> modulated_deform_conv_cuda_forward
 * (deform_conv_cuda.cpp:490-569).  Returns 0, or a negative code for shapes
 * the reference's shape_check (:62-150) rejects. */
int oracle_deform_conv_forward(const float *x, const float *offset,
                               const float *mask, const float *weight,
                               const float *bias, float *out, int B, int Cin,
                               int H, int W, int Cout, int kH, int kW, int sH,
                               int sW, int pH, int pW, int dH, int dW,
                               int groups, int dgroups, int acc64,
                               int nthreads) {
  if (kH <= 0 || kW <= 0 || sH <= 0 || sW <= 0 || dH <= 0 || dW <= 0) return -1;
  if (groups <= 0 || dgroups <= 0 || Cin % groups || Cout % groups ||
      Cin % dgroups)
    return -1;
  const int Hout = oracle_dcn_out_size(H, pH, dH, kH, sH);
  const int Wout = oracle_dcn_out_size(W, pW, dW, kW, sW);
  if (Hout < 1 || Wout < 1) return -2;
  if (!mask && (H < kH || W < kW)) return -3; /* deform_conv_cuda.cpp:127 */
  const int HWo = Hout * Wout;
  const size_t col_elems = (size_t)Cin * kH * kW * HWo;
  int fail = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1)
#endif
  for (int b = 0; b < B; ++b) {
    float *col = (float *)malloc(sizeof(float) * col_elems);
    if (!col) {
      fail = 1;
      continue;
    }
    oracle_im2col_one(x + (size_t)b * Cin * H * W,
                      offset + (size_t)b * dgroups * 2 * kH * kW * HWo,
                      mask ? mask + (size_t)b * dgroups * kH * kW * HWo : NULL,
                      Cin, H, W, kH, kW, pH, pW, sH, sW, dH, dW, dgroups, Hout,
                      Wout, col);
    oracle_gemm_one(weight, col, bias, Cin, Cout, kH * kW, groups, HWo, acc64,
                    out + (size_t)b * Cout * HWo);
    free(col);
  }
  (void)nthreads;
  return fail ? -4 : 0;
}

/* ------------------------------------------------------------------------
 * Backward of the deformable convolution (training):
 *   deform_conv_backward_input_cuda / deform_conv_backward_parameters_cuda
 *     (deform_conv_cuda.cpp:262-488) and modulated_deform_conv_cuda_backward
 *   kernels deformable_col2im (deform_conv_cuda_kernel.cu:279-330),
 *     deformable_col2im_coord (:363-427), their modulated forms (:636-755),
 *   weights get_gradient_weight (:117-144), get_coordinate_weight (:146-188).
 * Sums are taken in double (the checker should be the better value; the CUDA
 * code's float atomics have no defined order anyway).
 * ---------------------------------------------------------------------- */

static double oracle_gradient_weight(float ah, float aw, int h, int w, int H, int W) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  const int hl = (int)floorf(ah), wl = (int)floorf(aw);
  const int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
  return weight;
}

static float oracle_coordinate_weight(float ah, float aw, int H, int W, const float *im,
                                      int dir) {
  if (ah <= -1 || ah >= H || aw <= -1 || aw >= W) return 0;
  const int hl = (int)floorf(ah), wl = (int)floorf(aw);
  const int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * im[(size_t)hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += -1 * (aw - wl) * im[(size_t)hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += (wl + 1 - aw) * im[(size_t)hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (aw - wl) * im[(size_t)hh * W + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * im[(size_t)hl * W + wl];
    if (hl >= 0 && wh <= W - 1) weight += (hl + 1 - ah) * im[(size_t)hl * W + wh];
    if (hh <= H - 1 && wl >= 0) weight += -1 * (ah - hl) * im[(size_t)hh * W + wl];
    if (hh <= H - 1 && wh <= W - 1) weight += (ah - hl) * im[(size_t)hh * W + wh];
  }
  return weight;
}

/* All gradients of one call, NCHW float32.  mask NULL -> DCNv1 (grad_mask and grad_bias
 * ignored).  Every output is fully overwritten.  Returns 0 or a negative code. */
int oracle_deform_conv_backward(const float *x, const float *offset, const float *mask,
                                const float *weight, const float *grad_out,
                                float *grad_x, float *grad_offset, float *grad_mask,
                                float *grad_weight, float *grad_bias, int B, int Cin,
                                int H, int W, int Cout, int kH, int kW, int sH, int sW,
                                int pH, int pW, int dH, int dW, int groups, int dgroups) {
  if (kH <= 0 || kW <= 0 || sH <= 0 || sW <= 0 || dH <= 0 || dW <= 0) return -1;
  if (groups <= 0 || dgroups <= 0 || Cin % groups || Cout % groups || Cin % dgroups)
    return -1;
  const int Hout = oracle_dcn_out_size(H, pH, dH, kH, sH);
  const int Wout = oracle_dcn_out_size(W, pW, dW, kW, sW);
  if (Hout < 1 || Wout < 1) return -2;
  const int HWo = Hout * Wout, T = kH * kW;
  const int Cig = Cin / groups, Cog = Cout / groups, cpg = Cin / dgroups;
  const size_t nx = (size_t)B * Cin * H * W, nw = (size_t)Cout * Cig * T;
  double *gx = (double *)calloc(nx, sizeof(double));
  double *gw = (double *)calloc(nw, sizeof(double));
  double *gcol = (double *)malloc(sizeof(double) * (size_t)Cin * T * HWo);
  float *col = (float *)malloc(sizeof(float) * (size_t)Cin * T * HWo);
  if (!gx || !gw || !gcol || !col) return -4;
  for (int b = 0; b < B; ++b) {
    const float *go = grad_out + (size_t)b * Cout * HWo;
    const float *xb = x + (size_t)b * Cin * H * W;
    const float *off = offset + (size_t)b * dgroups * 2 * T * HWo;
    const float *msk = mask ? mask + (size_t)b * dgroups * T * HWo : NULL;
    /* columns = weight^T * gradOutput  (deform_conv_cuda.cpp:336-345) */
    for (int c = 0; c < Cin; ++c) {
      const int g = c / Cig, cl = c - g * Cig;
      for (int t = 0; t < T; ++t) {
        double *row = gcol + ((size_t)c * T + t) * HWo;
        for (int p = 0; p < HWo; ++p) row[p] = 0;
        for (int o = g * Cog; o < (g + 1) * Cog; ++o) {
          const double wv = weight[((size_t)o * Cig + cl) * T + t];
          const float *gr = go + (size_t)o * HWo;
          for (int p = 0; p < HWo; ++p) row[p] += wv * (double)gr[p];
        }
      }
    }
    /* the unmasked samples, for grad_weight and grad_mask */
    oracle_im2col_one(xb, off, NULL, Cin, H, W, kH, kW, pH, pW, sH, sW, dH, dW, dgroups,
                      Hout, Wout, col);
    for (int c = 0; c < Cin; ++c) {
      const int dg = c / cpg;
      for (int t = 0; t < T; ++t) {
        const int ki = t / kW, kj = t - ki * kW;
        for (int ho = 0; ho < Hout; ++ho)
          for (int wo = 0; wo < Wout; ++wo) {
            const int p = ho * Wout + wo;
            const size_t oh = ((size_t)(dg * 2 * T + 2 * t) * Hout + ho) * Wout + wo;
            const size_t ow = ((size_t)(dg * 2 * T + 2 * t + 1) * Hout + ho) * Wout + wo;
            const size_t om = ((size_t)(dg * T + t) * Hout + ho) * Wout + wo;
            const float m = msk ? msk[om] : 1.f;
            const double g = gcol[((size_t)c * T + t) * HWo + p];
            const float ih = (float)(ho * sH - pH + ki * dH) + off[oh];
            const float iw = (float)(wo * sW - pW + kj * dW) + off[ow];
            /* col2im (:279-330): the up-to-four cells within distance 1 */
            const int ch = (int)ih, cw = (int)iw;
            for (int dy = -2; dy <= 2; ++dy)
              for (int dx = -2; dx <= 2; ++dx) {
                const int yy = ch + dy, xx = cw + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && fabsf(ih - yy) < 1 &&
                    fabsf(iw - xx) < 1)
                  gx[((size_t)b * Cin + c) * H * W + (size_t)yy * W + xx] +=
                      oracle_gradient_weight(ih, iw, yy, xx, H, W) * g * (double)m;
              }
          }
      }
    }
    /* grad_offset / grad_mask: per (dg, t, p) sums over the group's channels */
    for (int dg = 0; dg < dgroups; ++dg)
      for (int t = 0; t < T; ++t) {
        const int ki = t / kW, kj = t - ki * kW;
        for (int ho = 0; ho < Hout; ++ho)
          for (int wo = 0; wo < Wout; ++wo) {
            const int p = ho * Wout + wo;
            const size_t oh = ((size_t)(dg * 2 * T + 2 * t) * Hout + ho) * Wout + wo;
            const size_t ow = ((size_t)(dg * 2 * T + 2 * t + 1) * Hout + ho) * Wout + wo;
            const size_t om = ((size_t)(dg * T + t) * Hout + ho) * Wout + wo;
            const float m = msk ? msk[om] : 1.f;
            float ih = (float)(ho * sH - pH + ki * dH) + off[oh];
            float iw = (float)(wo * sW - pW + kj * dW) + off[ow];
            const int inside = !(ih <= -1 || iw <= -1 || ih >= H || iw >= W);
            if (!inside) ih = iw = -2;
            double vh = 0, vw = 0, vm = 0;
            for (int c = dg * cpg; c < (dg + 1) * cpg; ++c) {
              const float *im = xb + (size_t)c * H * W;
              const double g = gcol[((size_t)c * T + t) * HWo + p];
              vh += (double)oracle_coordinate_weight(ih, iw, H, W, im, 0) * g * (double)m;
              vw += (double)oracle_coordinate_weight(ih, iw, H, W, im, 1) * g * (double)m;
              if (inside) vm += g * (double)col[((size_t)c * T + t) * HWo + p];
            }
            grad_offset[(size_t)b * dgroups * 2 * T * HWo + oh] = (float)vh;
            grad_offset[(size_t)b * dgroups * 2 * T * HWo + ow] = (float)vw;
            if (mask && grad_mask) grad_mask[(size_t)b * dgroups * T * HWo + om] = (float)vm;
          }
      }
    /* gradWeight += gradOutput * columns^T  (deform_conv_cuda.cpp:466-475) */
    for (int o = 0; o < Cout; ++o) {
      const int g = o / Cog;
      const float *gr = go + (size_t)o * HWo;
      for (int cl = 0; cl < Cig; ++cl) {
        const int c = g * Cig + cl;
        const int dg = c / cpg;
        for (int t = 0; t < T; ++t) {
          const float *cr = col + ((size_t)c * T + t) * HWo;
          const float *mr = msk ? msk + (size_t)(dg * T + t) * HWo : NULL;
          double acc = 0;
          for (int p = 0; p < HWo; ++p)
            acc += (double)gr[p] * (double)cr[p] * (mr ? (double)mr[p] : 1.0);
          gw[((size_t)o * Cig + cl) * T + t] += acc;
        }
      }
    }
  }
  for (size_t i = 0; i < nx; ++i) grad_x[i] = (float)gx[i];
  for (size_t i = 0; i < nw; ++i) grad_weight[i] = (float)gw[i];
  if (mask && grad_bias)
    for (int o = 0; o < Cout; ++o) {
      double acc = 0;
      for (int b = 0; b < B; ++b)
        for (int p = 0; p < HWo; ++p) acc += (double)grad_out[((size_t)b * Cout + o) * HWo + p];
      grad_bias[o] = (float)acc;
    }
  free(gx);
  free(gw);
  free(gcol);
  free(col);
  return 0;
}

/* ------------------------------------------------------------------------
 * Keypoint decode: GANetHeadFast.ktdet_decode_fast
 * (mmdet/models/dense_heads/ganet_head.py:203-238) for ONE image.
 * ---------------------------------------------------------------------- */

/* heat * (max_pool2d(heat,(1,3),1,(0,1)) == heat)   (ganet_head.py:205-208).
 * ATen's CPU max-pool starts from -inf and takes `val` when
 * (val > max) || isnan(val), so a NaN in the window wins.
 * nms_rows = 3 is the north star's 3x3 variant (max_pool2d(heat,(3,3),1,(1,1)),
 * window walked row-major as ATen does); the reference always runs 1x3. */
static float oracle_heat_nms(const float *heat, int H, int W, int y, int x,
                             int nms_rows) {
  float m = -INFINITY;
  const int r = nms_rows == 3 ? 1 : 0;
  for (int yy = y - r; yy <= y + r; ++yy) {
    if (yy < 0 || yy >= H) continue;
    for (int xx = x - 1; xx <= x + 1; ++xx) {
      if (xx < 0 || xx >= W) continue;
      float v = heat[(size_t)yy * W + xx];
      if (v > m || isnan(v)) m = v;
    }
  }
  float h = heat[(size_t)y * W + x];
  float keep = (m == h) ? 1.0f : 0.0f;
  return h * keep;
}

/* Outputs (caller allocates to capacity cap_c / cap_k, true counts are always
 * returned so the caller can detect overflow):
 *   centers[n_c][2]  int64 (x, y), x-major order   (ganet_head.py:213-219)
 *   kpt_xy[n_k][2]   int32 (x, y), row-major order (ganet_head.py:231)
 *   align[n_k][2]    float (x+err_x, y+err_y)      (ganet_head.py:230,233)
 *   root[n_k][2]     float (x+off_x, y+off_y)      (ganet_head.py:229,232)
 * thr and root_thr are compared in float32, as torch does for a Python
 * scalar against a float32 tensor. */
int oracle_decode_opt(const float *heat, const float *offset, const float *error,
                      int H, int W, float thr, float root_thr, int nms_rows,
                      int64_t *centers, int cap_c, int *n_c, int32_t *kpt_xy,
                      float *align, float *root, int cap_k, int *n_k) {
  const size_t HW = (size_t)H * W;
  int nc = 0, nk = 0;
  /* centres: mask transposed to [W,H] before np.where -> x-major */
  for (int x = 0; x < W; ++x)
    for (int y = 0; y < H; ++y) {
      float v = oracle_heat_nms(heat, H, W, y, x, nms_rows);
      if ((offset[HW + (size_t)y * W + x] < root_thr) && (v > thr)) {
        if (nc < cap_c) {
          centers[2 * nc] = x;
          centers[2 * nc + 1] = y;
        }
        ++nc;
      }
    }
  /* keypoints: torch.where on [H,W,2] -> row-major */
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      float v = oracle_heat_nms(heat, H, W, y, x, nms_rows);
      if (v > thr) {
        if (nk < cap_k) {
          size_t i = (size_t)y * W + x;
          kpt_xy[2 * nk] = x;
          kpt_xy[2 * nk + 1] = y;
          root[2 * nk] = (float)x + offset[i];
          root[2 * nk + 1] = (float)y + offset[HW + i];
          align[2 * nk] = (float)x + error[i];
          align[2 * nk + 1] = (float)y + error[HW + i];
        }
        ++nk;
      }
    }
  *n_c = nc;
  *n_k = nk;
  return (nc > cap_c || nk > cap_k) ? 1 : 0;
}

/* the reference's decode: 1x3 NMS */
int oracle_decode(const float *heat, const float *offset, const float *error,
                  int H, int W, float thr, float root_thr, int64_t *centers,
                  int cap_c, int *n_c, int32_t *kpt_xy, float *align,
                  float *root, int cap_k, int *n_k) {
  return oracle_decode_opt(heat, offset, error, H, W, thr, root_thr, 1, centers,
                           cap_c, n_c, kpt_xy, align, root, cap_k, n_k);
}

/* ------------------------------------------------------------------------
 * Lane grouping: group_points_fast (tools/ganet/post_process.py:141-194)
 * ---------------------------------------------------------------------- */

/* cal_dis (post_process.py:44-46) on two int64 points: integer squares,
 * math.sqrt of the exact integer. */
static double oracle_dis_ii(int64_t ax, int64_t ay, int64_t bx, int64_t by) {
  int64_t dx = ax - bx, dy = ay - by;
  return sqrt((double)(dx * dx + dy * dy));
}

/* cal_dis on float64 operands: numpy's scalar `**` is libm pow(). */
static double oracle_dis_dd(double ax, double ay, double bx, double by) {
  return sqrt(pow(ax - bx, 2.0) + pow(ay - by, 2.0));
}

/* by_center_thr >= 0: update_coords_fast_by_center (post_process.py:155-184),
 *   both thresholds = by_center_thr as the caller passes them (:190-193).
 * by_center_thr < 0 (None): update_coords_fast (:142-153) with `thr`.
 * Outputs:
 *   center_gid[nc]        group index of every centre (by-centre mode)
 *   group_mean[n_g][2]    int64 truncated mean (choose_mean_point :38-41)
 *   kpt_gid[nk]           group index of every keypoint, -1 = dropped
 * Returns the number of groups. */
int oracle_group_fast_opt(const int64_t *centers, int nc, const float *root,
                          int nk, double thr, double by_center_thr, int nearest,
                          int32_t *center_gid, int64_t *group_mean,
                          int32_t *kpt_gid) {
  int ng = 0;
  if (by_center_thr >= 0) {
    const double T = by_center_thr;
    /* search_groups (:49-58) returns the first group, in group order, that
     * holds ANY member within thr: that is the smallest group index among
     * the earlier points within thr. */
    for (int i = 0; i < nc; ++i) {
      int found = -1;
      for (int j = 0; j < i; ++j)
        if ((found < 0 || center_gid[j] < found) &&
            oracle_dis_ii(centers[2 * i], centers[2 * i + 1], centers[2 * j],
                          centers[2 * j + 1]) <= T)
          found = center_gid[j];
      if (found < 0) found = ng++;
      center_gid[i] = found;
    }
    for (int g = 0; g < ng; ++g) {
      int64_t sx = 0, sy = 0, cnt = 0;
      for (int j = 0; j < nc; ++j)
        if (center_gid[j] == g) {
          sx += centers[2 * j];
          sy += centers[2 * j + 1];
          ++cnt;
        }
      /* np.mean(..., dtype=int): int64 sum, true divide, unsafe cast */
      group_mean[2 * g] = (int64_t)((double)sx / (double)cnt);
      group_mean[2 * g + 1] = (int64_t)((double)sy / (double)cnt);
    }
    /* search_groups_by_centers (:61-66): first centre within thr.
     * nearest != 0 is the north star's variant: the NEAREST centre within thr
     * (ties go to the lower group index); the reference always takes the first. */
    for (int i = 0; i < nk; ++i) {
      int found = -1;
      double best = 0;
      for (int g = 0; g < ng; ++g) {
        const double d = oracle_dis_dd((double)root[2 * i], (double)root[2 * i + 1],
                                       (double)group_mean[2 * g],
                                       (double)group_mean[2 * g + 1]);
        if (d <= T && (found < 0 || d < best)) {
          found = g;
          best = d;
          if (!nearest) break;
        }
      }
      kpt_gid[i] = found;
    }
  } else {
    for (int i = 0; i < nk; ++i) {
      int found = -1;
      for (int j = 0; j < i; ++j)
        if ((found < 0 || kpt_gid[j] < found) &&
            oracle_dis_dd((double)root[2 * i], (double)root[2 * i + 1],
                          (double)root[2 * j], (double)root[2 * j + 1]) <= thr)
          found = kpt_gid[j];
      if (found < 0) found = ng++;
      kpt_gid[i] = found;
    }
  }
  return ng;
}

/* the reference's grouping: first centre within thr */
int oracle_group_fast(const int64_t *centers, int nc, const float *root,
                      int nk, double thr, double by_center_thr,
                      int32_t *center_gid, int64_t *group_mean,
                      int32_t *kpt_gid) {
  return oracle_group_fast_opt(centers, nc, root, nk, thr, by_center_thr, 0,
                               center_gid, group_mean, kpt_gid);
}

/* Whole per-image decode + grouping over a batch, for the CPU baseline:
 * images are independent, so OpenMP over images is the natural CPU port.
 * Returns total number of grouped keypoints (a checksum the caller can use). */
long oracle_decode_group_batch(const float *heat, const float *offset,
                               const float *error, int B, int H, int W,
                               float thr, float root_thr, double cluster_thr,
                               double by_center_thr, int cap, int nthreads) {
  const size_t HW = (size_t)H * W;
  long total = 0;
#ifdef _OPENMP
  if (nthreads < 1) nthreads = 1;
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 1) reduction(+ : total)
#endif
  for (int b = 0; b < B; ++b) {
    int64_t *centers = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)cap);
    int32_t *kxy = (int32_t *)malloc(sizeof(int32_t) * 2 * (size_t)cap);
    float *al = (float *)malloc(sizeof(float) * 2 * (size_t)cap);
    float *rt = (float *)malloc(sizeof(float) * 2 * (size_t)cap);
    int32_t *cg = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    int32_t *kg = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    int64_t *gm = (int64_t *)malloc(sizeof(int64_t) * 2 * (size_t)cap);
    int nc = 0, nk = 0;
    oracle_decode(heat + b * HW, offset + 2 * b * HW, error + 2 * b * HW, H, W,
                  thr, root_thr, centers, cap, &nc, kxy, al, rt, cap, &nk);
    if (nc > cap) nc = cap;
    if (nk > cap) nk = cap;
    oracle_group_fast(centers, nc, rt, nk, cluster_thr, by_center_thr, cg, gm,
                      kg);
    for (int i = 0; i < nk; ++i) total += (kg[i] >= 0);
    free(centers); free(kxy); free(al); free(rt); free(cg); free(kg); free(gm);
  }
  (void)nthreads;
  return total;
}
