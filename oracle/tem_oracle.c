/*
 * tem_oracle.c -- CPU restatement of the convolution arithmetic of transfer_em's
 * CycleGAN hot path.  TEST INFRASTRUCTURE ONLY: nothing under transfer_em_amd/
 * may link, load or call this file; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg do.
 *
 * PARITY UNPINNED: the reference delegates all arithmetic to TensorFlow 2 /
 * tensorflow_addons, neither of which is present in /root/reference nor
 * installable here, and the reference has no tests or golden vectors.  This file
 * restates the *published* semantics of the Keras layers the reference calls:
 *
 *   Conv2D/Conv3D            cross-correlation, channels-last, kernel laid out
 *                            (kd,kh,kw,C_in,C_out)      transfer_em/models/utils.py:73,80
 *                                                       transfer_em/models/generator.py:54,96,108,110
 *                                                       transfer_em/models/discriminator.py:45,78,97
 *   Conv2D/3DTranspose       stride 2, k 4, padding 'same': o = 2*j + t - 1,
 *                            kernel (kd,kh,kw,C_out,C_in) transfer_em/models/utils.py:129-130
 *
 * All tensors are dense NDHWC float32 (2-D is D == 1, kd == 1).  Accumulation is
 * in double so that the oracle is the better-conditioned side of every compare.
 * Loops are the plain definition of each operator; no tiling, no im2col.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#if defined(__GNUC__) && defined(__x86_64__)
#define ORC_CLONES __attribute__((target_clones("avx2,fma", "default")))
#else
#define ORC_CLONES
#endif

typedef struct {
  int N, D, H, W, C; /* NDHWC dims of the (dense) input tensor  */
} orc_shape;

/* out[n,oz,oy,ox,co] = bias[co] + sum_{dz,dy,dx,ci} in[n, oz*s+dz-p, oy*s+dy-p, ox*s+dx-p, ci]
 *                                                  * w[dz,dy,dx,ci,co]
 * stride s[3] / pad p[3] per axis (pad may be negative == crop); out-of-range
 * input taps read as zero.  Output dims are given by the caller. */
ORC_CLONES
void orc_conv_fwd(const float *in, int N, int D, int H, int W, int CI,
                  const float *w, int kd, int kh, int kw, int CO,
                  const int *s, const int *p, const float *bias,
                  float *out, int OD, int OH, int OW)
{
  #pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int oz = 0; oz < OD; ++oz)
      for (int oy = 0; oy < OH; ++oy) {
        double acc[64];
        for (int ox = 0; ox < OW; ++ox) {
          for (int co = 0; co < CO; ++co) acc[co] = bias ? (double)bias[co] : 0.0;
          for (int dz = 0; dz < kd; ++dz) {
            int iz = oz * s[0] + dz - p[0];
            if (iz < 0 || iz >= D) continue;
            for (int dy = 0; dy < kh; ++dy) {
              int iy = oy * s[1] + dy - p[1];
              if (iy < 0 || iy >= H) continue;
              for (int dx = 0; dx < kw; ++dx) {
                int ix = ox * s[2] + dx - p[2];
                if (ix < 0 || ix >= W) continue;
                const float *xp = in + ((((size_t)n * D + iz) * H + iy) * W + ix) * CI;
                const float *wp = w + (((size_t)dz * kh + dy) * kw + dx) * CI * CO;
                for (int ci = 0; ci < CI; ++ci) {
                  double xv = xp[ci];
                  const float *wr = wp + (size_t)ci * CO;
                  for (int co = 0; co < CO; ++co) acc[co] += xv * (double)wr[co];
                }
              }
            }
          }
          float *op = out + ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * CO;
          for (int co = 0; co < CO; ++co) op[co] = (float)acc[co];
        }
      }
}

/* Gradient of orc_conv_fwd w.r.t. its input:
 * din[n,iz,iy,ix,ci] = sum over (o,tap) with o*s + tap - p == i of dout[n,o,co] * w[tap,ci,co]. */
ORC_CLONES
void orc_conv_bwd_data(const float *dout, int N, int OD, int OH, int OW, int CO,
                       const float *w, int kd, int kh, int kw, int CI,
                       const int *s, const int *p,
                       float *din, int D, int H, int W)
{
  #pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int iz = 0; iz < D; ++iz)
      for (int iy = 0; iy < H; ++iy) {
        double acc[64];
        for (int ix = 0; ix < W; ++ix) {
          for (int ci = 0; ci < CI; ++ci) acc[ci] = 0.0;
          for (int dz = 0; dz < kd; ++dz) {
            int tz = iz + p[0] - dz;
            if (tz < 0 || tz % s[0]) continue;
            int oz = tz / s[0];
            if (oz >= OD) continue;
            for (int dy = 0; dy < kh; ++dy) {
              int ty = iy + p[1] - dy;
              if (ty < 0 || ty % s[1]) continue;
              int oy = ty / s[1];
              if (oy >= OH) continue;
              for (int dx = 0; dx < kw; ++dx) {
                int tx = ix + p[2] - dx;
                if (tx < 0 || tx % s[2]) continue;
                int ox = tx / s[2];
                if (ox >= OW) continue;
                const float *gp = dout + ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * CO;
                const float *wp = w + (((size_t)dz * kh + dy) * kw + dx) * CI * CO;
                for (int ci = 0; ci < CI; ++ci) {
                  const float *wr = wp + (size_t)ci * CO;
                  double a = 0.0;
                  for (int co = 0; co < CO; ++co) a += (double)gp[co] * (double)wr[co];
                  acc[ci] += a;
                }
              }
            }
          }
          float *dp = din + ((((size_t)n * D + iz) * H + iy) * W + ix) * CI;
          for (int ci = 0; ci < CI; ++ci) dp[ci] = (float)acc[ci];
        }
      }
}

/* Gradient of orc_conv_fwd w.r.t. the kernel (double output, caller rounds):
 * dw[tap,ci,co] = sum_{n,o} in[n, o*s+tap-p, ci] * dout[n,o,co]. */
ORC_CLONES
void orc_conv_bwd_weight(const float *in, int N, int D, int H, int W, int CI,
                         const float *dout, int OD, int OH, int OW, int CO,
                         int kd, int kh, int kw, const int *s, const int *p,
                         double *dw)
{
  int ntap = kd * kh * kw;
  #pragma omp parallel for schedule(dynamic, 1)
  for (int tap = 0; tap < ntap; ++tap) {
    int dz = tap / (kh * kw), dy = (tap / kw) % kh, dx = tap % kw;
    double *dwp = dw + (size_t)tap * CI * CO;
    for (size_t i = 0; i < (size_t)CI * CO; ++i) dwp[i] = 0.0;
    for (int n = 0; n < N; ++n)
      for (int oz = 0; oz < OD; ++oz) {
        int iz = oz * s[0] + dz - p[0];
        if (iz < 0 || iz >= D) continue;
        for (int oy = 0; oy < OH; ++oy) {
          int iy = oy * s[1] + dy - p[1];
          if (iy < 0 || iy >= H) continue;
          for (int ox = 0; ox < OW; ++ox) {
            int ix = ox * s[2] + dx - p[2];
            if (ix < 0 || ix >= W) continue;
            const float *xp = in + ((((size_t)n * D + iz) * H + iy) * W + ix) * CI;
            const float *gp = dout + ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * CO;
            for (int ci = 0; ci < CI; ++ci) {
              double xv = xp[ci];
              double *dr = dwp + (size_t)ci * CO;
              for (int co = 0; co < CO; ++co) dr[co] += xv * (double)gp[co];
            }
          }
        }
      }
  }
}

/* Keras Conv{2,3}DTranspose forward, kernel laid out (kd,kh,kw,C_out,C_in):
 * out[n,o,co] = sum_{j,t : o == j*s + t - p} in[n,j,ci] * w[t,co,ci].
 * 'same' padding with k = 4, s = 2 is p = 1, out = 2*in (models/utils.py:129-130). */
ORC_CLONES
void orc_convT_fwd(const float *in, int N, int D, int H, int W, int CI,
                   const float *w, int kd, int kh, int kw, int CO,
                   const int *s, const int *p,
                   float *out, int OD, int OH, int OW)
{
  #pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int oz = 0; oz < OD; ++oz)
      for (int oy = 0; oy < OH; ++oy) {
        double acc[64];
        for (int ox = 0; ox < OW; ++ox) {
          for (int co = 0; co < CO; ++co) acc[co] = 0.0;
          for (int dz = 0; dz < kd; ++dz) {
            int tz = oz + p[0] - dz;
            if (tz < 0 || tz % s[0]) continue;
            int jz = tz / s[0];
            if (jz >= D) continue;
            for (int dy = 0; dy < kh; ++dy) {
              int ty = oy + p[1] - dy;
              if (ty < 0 || ty % s[1]) continue;
              int jy = ty / s[1];
              if (jy >= H) continue;
              for (int dx = 0; dx < kw; ++dx) {
                int tx = ox + p[2] - dx;
                if (tx < 0 || tx % s[2]) continue;
                int jx = tx / s[2];
                if (jx >= W) continue;
                const float *xp = in + ((((size_t)n * D + jz) * H + jy) * W + jx) * CI;
                const float *wp = w + (((size_t)dz * kh + dy) * kw + dx) * CO * CI;
                for (int co = 0; co < CO; ++co) {
                  const float *wr = wp + (size_t)co * CI;
                  double a = 0.0;
                  for (int ci = 0; ci < CI; ++ci) a += (double)xp[ci] * (double)wr[ci];
                  acc[co] += a;
                }
              }
            }
          }
          float *op = out + ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * CO;
          for (int co = 0; co < CO; ++co) op[co] = (float)acc[co];
        }
      }
}

/* Gradient of orc_convT_fwd w.r.t. its input:
 * din[n,j,ci] = sum_{t,co} dout[n, j*s+t-p, co] * w[t,co,ci]   (a strided conv). */
ORC_CLONES
void orc_convT_bwd_data(const float *dout, int N, int OD, int OH, int OW, int CO,
                        const float *w, int kd, int kh, int kw, int CI,
                        const int *s, const int *p,
                        float *din, int D, int H, int W)
{
  #pragma omp parallel for collapse(3) schedule(static)
  for (int n = 0; n < N; ++n)
    for (int jz = 0; jz < D; ++jz)
      for (int jy = 0; jy < H; ++jy) {
        double acc[64];
        for (int jx = 0; jx < W; ++jx) {
          for (int ci = 0; ci < CI; ++ci) acc[ci] = 0.0;
          for (int dz = 0; dz < kd; ++dz) {
            int oz = jz * s[0] + dz - p[0];
            if (oz < 0 || oz >= OD) continue;
            for (int dy = 0; dy < kh; ++dy) {
              int oy = jy * s[1] + dy - p[1];
              if (oy < 0 || oy >= OH) continue;
              for (int dx = 0; dx < kw; ++dx) {
                int ox = jx * s[2] + dx - p[2];
                if (ox < 0 || ox >= OW) continue;
                const float *gp = dout + ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * CO;
                const float *wp = w + (((size_t)dz * kh + dy) * kw + dx) * CO * CI;
                for (int co = 0; co < CO; ++co) {
                  double g = gp[co];
                  const float *wr = wp + (size_t)co * CI;
                  for (int ci = 0; ci < CI; ++ci) acc[ci] += g * (double)wr[ci];
                }
              }
            }
          }
          float *dp = din + ((((size_t)n * D + jz) * H + jy) * W + jx) * CI;
          for (int ci = 0; ci < CI; ++ci) dp[ci] = (float)acc[ci];
        }
      }
}

/* Gradient of orc_convT_fwd w.r.t. the kernel:
 * dw[t,co,ci] = sum_{n,j} in[n,j,ci] * dout[n, j*s+t-p, co]. */
ORC_CLONES
void orc_convT_bwd_weight(const float *in, int N, int D, int H, int W, int CI,
                          const float *dout, int OD, int OH, int OW, int CO,
                          int kd, int kh, int kw, const int *s, const int *p,
                          double *dw)
{
  int ntap = kd * kh * kw;
  #pragma omp parallel for schedule(dynamic, 1)
  for (int tap = 0; tap < ntap; ++tap) {
    int dz = tap / (kh * kw), dy = (tap / kw) % kh, dx = tap % kw;
    double *dwp = dw + (size_t)tap * CO * CI;
    for (size_t i = 0; i < (size_t)CI * CO; ++i) dwp[i] = 0.0;
    for (int n = 0; n < N; ++n)
      for (int jz = 0; jz < D; ++jz) {
        int oz = jz * s[0] + dz - p[0];
        if (oz < 0 || oz >= OD) continue;
        for (int jy = 0; jy < H; ++jy) {
          int oy = jy * s[1] + dy - p[1];
          if (oy < 0 || oy >= OH) continue;
          for (int jx = 0; jx < W; ++jx) {
            int ox = jx * s[2] + dx - p[2];
            if (ox < 0 || ox >= OW) continue;
            const float *xp = in + ((((size_t)n * D + jz) * H + jy) * W + jx) * CI;
            const float *gp = dout + ((((size_t)n * OD + oz) * OH + oy) * OW + ox) * CO;
            for (int co = 0; co < CO; ++co) {
              double g = gp[co];
              double *dr = dwp + (size_t)co * CI;
              for (int ci = 0; ci < CI; ++ci) dr[ci] += g * (double)xp[ci];
            }
          }
        }
      }
  }
}

/* Philox4x32-10 (Salmon et al., SC'11; Random123 reference constants).  Used for
 * the dropout keep-mask so that the HIP epilogue and the oracle draw the same
 * bits.  Keras Dropout(0.5) itself (models/utils.py:134) only promises an
 * independent Bernoulli(keep = 0.5) per element with survivors scaled by 2. */
static inline void philox_round(uint32_t c[4], const uint32_t k[2])
{
  uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0];
  uint32_t n1 = (uint32_t)p1;
  uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1];
  uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  uint32_t k[2] = {key[0], key[1]};
  for (int r = 0; r < 10; ++r) {
    if (r) { k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u; }
    philox_round(c, k);
  }
  memcpy(out, c, sizeof c);
}

/* keep[e] = bit (e & 31) of word ((e >> 5) & 3) of
 * philox4x32_10(ctr = {lo32(e>>7), hi32(e>>7), site, step}, key = {seed_lo, seed_hi}). */
void orc_dropout_mask(uint8_t *keep, uint64_t count, uint64_t seed, uint32_t site, uint32_t step)
{
  uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
  #pragma omp parallel for schedule(static)
  for (uint64_t blk = 0; blk < (count + 127) / 128; ++blk) {
    uint32_t ctr[4] = {(uint32_t)blk, (uint32_t)(blk >> 32), site, step};
    uint32_t r[4];
    orc_philox4x32_10(ctr, key, r);
    uint64_t e0 = blk * 128, e1 = e0 + 128 < count ? e0 + 128 : count;
    for (uint64_t e = e0; e < e1; ++e)
      keep[e] = (uint8_t)((r[(e >> 5) & 3] >> (e & 31)) & 1u);
  }
}
