/*
 * unina_oracle.c -- CPU ORACLE, forward graph (test infrastructure, NOT product code).
 *
 * fp32 NCHW restatement of the reference forward, module by module, taking the
 * reference's UNFOLDED parameters (conv weight + BN gamma/beta/mean/var), so that the
 * engine's BN folding is itself under test.
 *
 * Reference: /root/reference/unina_yolo_dla/model.py
 *   ConvBlock      :23-50     Bottleneck :53-73    C3k2     :76-110
 *   SPPF_DLA       :113-132   Upsample   :135-147  Backbone :152-219
 *   Neck           :224-269   DetectionHead :274-303   UNINA_YOLO_DLA.forward :347-365
 *
 * Also used as bench.py's cpu_baseline ("port"): the convolution is a register-tiled
 * 4x16 GEMM micro-kernel over a zero-padded copy of the input (AVX2/FMA through GCC
 * vector extensions, OpenMP over output tiles) so that the denominator is a fair one.
 */
#define _GNU_SOURCE
#include "unina_oracle.h"

#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ errors */
static char g_err[512];
const char *uo_last_error(void) { return g_err; }
static void set_err(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

/* -------------------------------------------------------------- state dict */
typedef struct {
  char *name;
  int ndim;
  int dims[4];
  float *data;
} sd_entry;
struct uo_statedict {
  int count;
  sd_entry *e;
};

uo_statedict *uo_sd_load(const char *path) {
  FILE *f = fopen(path, "rb");
  if (!f) {
    set_err("cannot open %s", path);
    return NULL;
  }
  char magic[8];
  uint32_t count = 0;
  if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "UNSD0001", 8) || fread(&count, 4, 1, f) != 1) {
    set_err("%s: bad UNSD header", path);
    fclose(f);
    return NULL;
  }
  uo_statedict *sd = calloc(1, sizeof *sd);
  sd->count = (int)count;
  sd->e = calloc(count, sizeof(sd_entry));
  for (uint32_t i = 0; i < count; ++i) {
    uint16_t nl;
    uint32_t nd, dims[8];
    if (fread(&nl, 2, 1, f) != 1) goto bad;
    sd->e[i].name = calloc(nl + 1, 1);
    if (fread(sd->e[i].name, 1, nl, f) != nl) goto bad;
    if (fread(&nd, 4, 1, f) != 1 || nd > 4) goto bad;
    if (nd && fread(dims, 4, nd, f) != nd) goto bad;
    size_t n = 1;
    sd->e[i].ndim = (int)nd;
    for (uint32_t d = 0; d < nd; ++d) {
      sd->e[i].dims[d] = (int)dims[d];
      n *= dims[d];
    }
    sd->e[i].data = malloc(n * sizeof(float));
    if (fread(sd->e[i].data, 4, n, f) != n) goto bad;
  }
  fclose(f);
  return sd;
bad:
  set_err("%s: truncated UNSD file", path);
  fclose(f);
  uo_sd_free(sd);
  return NULL;
}

void uo_sd_free(uo_statedict *sd) {
  if (!sd) return;
  for (int i = 0; i < sd->count; ++i) {
    free(sd->e[i].name);
    free(sd->e[i].data);
  }
  free(sd->e);
  free(sd);
}
int uo_sd_count(const uo_statedict *sd) { return sd->count; }

const float *uo_sd_get(const uo_statedict *sd, const char *name, int *ndim, int dims[4]) {
  for (int i = 0; i < sd->count; ++i)
    if (!strcmp(sd->e[i].name, name)) {
      if (ndim) *ndim = sd->e[i].ndim;
      if (dims) memcpy(dims, sd->e[i].dims, sizeof sd->e[i].dims);
      return sd->e[i].data;
    }
  return NULL;
}

/* ------------------------------------------------------------------ tensors */
typedef struct {
  int c, h, w;
  float *d; /* [c][h][w] */
} ten;

/* One persistent slab, bump-allocated per forward; pages stay mapped between calls so the
 * cpu_baseline timing is not dominated by page faults. One live run at a time. */
static char *g_slab;
static size_t g_slab_cap, g_slab_used;
static int g_slab_live;

static float *slab_alloc(size_t nfloats) {
  size_t bytes = (nfloats * sizeof(float) + 63) & ~(size_t)63;
  if (g_slab_used + bytes > g_slab_cap) return NULL;
  float *p = (float *)(g_slab + g_slab_used);
  g_slab_used += bytes;
  return p;
}

#define MAX_NAMED 256
struct uo_run {
  int n;
  char *names[MAX_NAMED];
  ten t[MAX_NAMED];
  int keep_all;
  int failed;
};

static ten new_ten(uo_run *r, int c, int h, int w) {
  ten t = {c, h, w, slab_alloc((size_t)c * h * w)};
  if (!t.d) {
    r->failed = 1;
    set_err("oracle slab exhausted");
  }
  return t;
}
static void keep(uo_run *r, const char *name, ten t, int always) {
  if (!(r->keep_all || always) || r->n >= MAX_NAMED) return;
  r->names[r->n] = strdup(name);
  r->t[r->n++] = t;
}

/* ------------------------------------------------------------- convolution */
/* Register-tiled GEMM micro-kernels over GCC vector extensions: MR output channels x NV vectors of output pixels
 * (one row segment), k-sequential fp32 FMA accumulation per output -- so every variant below produces the same bits.
 * AVX2 (the baseline the .so is built for, x86-64-v3): 4 x 16 pixels. AVX-512 (picked at run time when the host has
 * it): 8 x 32 and 8 x 16 pixels. */
typedef float v8f __attribute__((vector_size(32)));
typedef float v16f __attribute__((vector_size(64)));
#define KC 256
#define MC 64 /* output channels per work unit: its weight block (MC x K floats) is streamed once per pixel-row unit */

#define SPLAT8(x) {x, x, x, x, x, x, x, x}
#define SPLAT16(x) {x, x, x, x, x, x, x, x, x, x, x, x, x, x, x, x}
#define DEF_MICRO(NAME, VT, VL, MR_, NV_, ATTR, SPLAT)                                                                    \
  ATTR static void NAME(const float *restrict wp, const float *restrict base, const int64_t *restrict koff, int kn, \
                        float *restrict acc) {                                                                      \
    VT a[MR_][NV_];                                                                                                 \
    _Pragma("GCC unroll 16") for (int m = 0; m < MR_; ++m)                                                         \
        _Pragma("GCC unroll 4") for (int v = 0; v < NV_; ++v) memcpy(&a[m][v], acc + (m * NV_ + v) * VL, sizeof(VT)); \
    for (int k = 0; k < kn; ++k) {                                                                                  \
      const float *bp = base + koff[k];                                                                             \
      const float *w = wp + (size_t)k * MR_;                                                                        \
      VT b[NV_];                                                                                                    \
      _Pragma("GCC unroll 4") for (int v = 0; v < NV_; ++v) memcpy(&b[v], bp + v * VL, sizeof(VT));                 \
      _Pragma("GCC unroll 16") for (int m = 0; m < MR_; ++m) {                                                     \
        const VT wv = SPLAT(w[m]);                                                                                  \
        _Pragma("GCC unroll 4") for (int v = 0; v < NV_; ++v) a[m][v] += wv * b[v];                                 \
      }                                                                                                             \
    }                                                                                                               \
    _Pragma("GCC unroll 16") for (int m = 0; m < MR_; ++m)                                                         \
        _Pragma("GCC unroll 4") for (int v = 0; v < NV_; ++v) memcpy(acc + (m * NV_ + v) * VL, &a[m][v], sizeof(VT)); \
  }
#define T_AVX512 __attribute__((target("avx512f,avx512vl,fma")))
DEF_MICRO(micro_avx2_4x16, v8f, 8, 4, 2, , SPLAT8)
DEF_MICRO(micro_avx512_8x32, v16f, 16, 8, 2, T_AVX512, SPLAT16)
DEF_MICRO(micro_avx512_8x16, v16f, 16, 8, 1, T_AVX512, SPLAT16)

typedef void (*micro_fn)(const float *, const float *, const int64_t *, int, float *);

/* scratch that lives across calls (grown on demand): a per-call calloc of the padded planes page-faults on every layer */
static float *g_src, *g_wp;
static int64_t *g_koff;
static size_t g_src_cap, g_wp_cap, g_koff_cap;
static int grow(void **p, size_t *cap, size_t bytes) {
  if (bytes <= *cap) return 0;
  free(*p);
  *p = NULL;
  if (posix_memalign(p, 64, bytes + 4096)) {
    *cap = 0;
    return -1;
  }
  memset(*p, 0, bytes + 4096);
  *cap = bytes;
  return 0;
}

/* Conv2d, cross-correlation, zero padding k/2, stride s in {1,2}, k in {1,3}, optional bias
 * (model.py:41-44 uses bias=False; the head output convs model.py:292,299 use bias=True).
 * wgt is [O][C][k][k] exactly as in the reference state_dict. */
static ten conv2d(uo_run *r, ten in, const float *wgt, const float *bias, int O, int k, int s) {
  const int C = in.c, H = in.h, W = in.w, p = k / 2;
  const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  ten out = new_ten(r, O, Ho, Wo);
  if (r->failed) return out;
  const int K = C * k * k;

  /* micro-kernel for this host and this row width */
  int MR = 4, NR = 16;
  micro_fn micro = micro_avx2_4x16;
  if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512vl")) {
    MR = 8;
    if (Wo % 32 == 0 || Wo >= 96) {
      NR = 32;
      micro = micro_avx512_8x32;
    } else {
      micro = micro_avx512_8x16;
    }
  }

  /* 1. source planes with contiguous x for every tap (zero padded; + NR floats of slack for the row tails) */
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  const int Hh = (Hp + 1) / 2, Wh = (Wp + 1) / 2;
  const size_t plane = (size_t)C * Hh * Wh;
  const size_t src_floats = (s == 1 ? (size_t)C * Hp * Wp : 4 * plane) + 64;
  const int panels = (O + MR - 1) / MR;
  if (grow((void **)&g_src, &g_src_cap, src_floats * sizeof(float)) || grow((void **)&g_koff, &g_koff_cap, sizeof(int64_t) * (size_t)K) ||
      grow((void **)&g_wp, &g_wp_cap, sizeof(float) * (size_t)panels * K * MR)) {
    r->failed = 1;
    set_err("oracle: conv scratch allocation failed");
    return out;
  }
  float *src = g_src;
  int64_t *koff = g_koff;
  float *wp = g_wp;
  int64_t rs; /* row stride of the planes */
  if (s == 1) {
    rs = Wp;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
      float *pl = src + (size_t)c * Hp * Wp;
      if (p) {
        memset(pl, 0, sizeof(float) * (size_t)Wp * p);
        memset(pl + (size_t)(Hp - p) * Wp, 0, sizeof(float) * (size_t)Wp * p);
      }
      for (int y = 0; y < H; ++y) {
        float *row = pl + (size_t)(y + p) * Wp;
        for (int j = 0; j < p; ++j) row[j] = 0.0f, row[Wp - 1 - j] = 0.0f;
        memcpy(row + p, in.d + ((size_t)c * H + y) * W, sizeof(float) * W);
      }
    }
    memset(src + (size_t)C * Hp * Wp, 0, 64 * sizeof(float));
    for (int c = 0, kk = 0; c < C; ++c)
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw, ++kk) koff[kk] = ((int64_t)c * Hp + kh) * Wp + kw;
  } else {
    /* stride 2: split the padded input into 4 parity planes Q[a][b][c][yy][xx] = pad[c][2yy+a][2xx+b] */
    rs = Wh;
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
      for (int q = 0; q < 4; ++q) memset(src + (size_t)q * plane + (size_t)c * Hh * Wh, 0, sizeof(float) * (size_t)Hh * Wh);
      for (int y = 0; y < H; ++y) {
        const int py = y + p;
        for (int x = 0; x < W; ++x) {
          const int px = x + p;
          src[(size_t)((py & 1) * 2 + (px & 1)) * plane + ((size_t)c * Hh + (py >> 1)) * Wh + (px >> 1)] =
              in.d[((size_t)c * H + y) * W + x];
        }
      }
    }
    memset(src + 4 * plane, 0, 64 * sizeof(float));
    for (int c = 0, kk = 0; c < C; ++c)
      for (int kh = 0; kh < k; ++kh)
        for (int kw = 0; kw < k; ++kw, ++kk)
          koff[kk] = (int64_t)((kh & 1) * 2 + (kw & 1)) * (int64_t)plane + ((int64_t)c * Hh + (kh >> 1)) * Wh + (kw >> 1);
  }

  /* 2. weights packed per MR-panel: wp[panel][K][MR], zero-padded rows */
#pragma omp parallel for schedule(static)
  for (int pn = 0; pn < panels; ++pn)
    for (int kk = 0; kk < K; ++kk)
      for (int m = 0; m < MR; ++m) {
        const int o = pn * MR + m;
        wp[((size_t)pn * K + kk) * MR + m] = o < O ? wgt[(size_t)o * K + kk] : 0.0f;
      }

  /* 3. work units: (block of MC channels, output row); the row's pixel tiles reuse the block's weights from cache */
  const int xt = (Wo + NR - 1) / NR;
  const int ppb = MC / MR; /* panels per channel block */
  const int cblocks = (panels + ppb - 1) / ppb;
#pragma omp parallel
  {
    float tmp[MC * 32] __attribute__((aligned(64)));
#pragma omp for collapse(2) schedule(dynamic, 1)
    for (int cb = 0; cb < cblocks; ++cb)
      for (int y = 0; y < Ho; ++y) {
        const int pn0 = cb * ppb, pn1 = pn0 + ppb < panels ? pn0 + ppb : panels;
        for (int xi = 0; xi < xt; ++xi) {
          const int x0 = xi * NR;
          const float *base = src + (int64_t)y * rs + x0;
          memset(tmp, 0, sizeof(float) * (size_t)(pn1 - pn0) * MR * NR);
          for (int k0 = 0; k0 < K; k0 += KC) {
            const int kn = K - k0 < KC ? K - k0 : KC;
            for (int pn = pn0; pn < pn1; ++pn)
              micro(wp + ((size_t)pn * K + k0) * MR, base, koff + k0, kn, tmp + (size_t)(pn - pn0) * MR * NR);
          }
          const int nx = Wo - x0 < NR ? Wo - x0 : NR;
          const int o1 = pn1 * MR < O ? pn1 * MR : O;
          for (int o = pn0 * MR; o < o1; ++o) {
            const float b = bias ? bias[o] : 0.0f;
            float *dst = out.d + ((size_t)o * Ho + y) * Wo + x0;
            const float *t = tmp + (size_t)(o - pn0 * MR) * NR;
            for (int j = 0; j < nx; ++j) dst[j] = t[j] + b;
          }
        }
      }
  }
  return out;
}

/* BatchNorm2d (eval) + ReLU, in place: y = (x-mean)/sqrt(var+eps)*gamma+beta  (model.py:46-50) */
static void bn_relu(ten t, const float *gamma, const float *beta, const float *mean, const float *var) {
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < t.c; ++c)
    for (int y = 0; y < t.h; ++y) {
      const float invstd = 1.0f / sqrtf(var[c] + 1e-5f);
      float *d = t.d + ((size_t)c * t.h + y) * t.w;
      for (int i = 0; i < t.w; ++i) {
        float v = (d[i] - mean[c]) * invstd * gamma[c] + beta[c];
        d[i] = v > 0.0f ? v : 0.0f;
      }
    }
}

/* ------------------------------------------------------------------ modules */
typedef struct {
  uo_run *r;
  const uo_statedict *sd;
} net;

static const float *param(net *n, const char *mod, const char *suffix, int expect) {
  char key[256];
  snprintf(key, sizeof key, "%s.%s", mod, suffix);
  int nd, dims[4];
  const float *p = uo_sd_get(n->sd, key, &nd, dims);
  if (!p) {
    n->r->failed = 1;
    set_err("missing parameter %s", key);
    return NULL;
  }
  size_t cnt = 1;
  for (int i = 0; i < nd; ++i) cnt *= (size_t)dims[i];
  if ((int)cnt != expect) {
    n->r->failed = 1;
    set_err("parameter %s has %zu elements, expected %d", key, cnt, expect);
    return NULL;
  }
  return p;
}

/* ConvBlock (model.py:23-50) */
static ten conv_block(net *n, const char *name, ten x, int cout, int k, int s) {
  ten bad = {0, 0, 0, NULL};
  const float *w = param(n, name, "conv.weight", cout * x.c * k * k);
  const float *g = param(n, name, "bn.weight", cout), *b = param(n, name, "bn.bias", cout);
  const float *m = param(n, name, "bn.running_mean", cout), *v = param(n, name, "bn.running_var", cout);
  if (n->r->failed) return bad;
  ten y = conv2d(n->r, x, w, NULL, cout, k, s);
  if (n->r->failed) return bad;
  bn_relu(y, g, b, m, v);
  keep(n->r, name, y, 0);
  return y;
}

static ten add(net *n, const char *name, ten a, ten b) {
  ten y = new_ten(n->r, a.c, a.h, a.w);
  if (n->r->failed) return y;
  const size_t hw = (size_t)a.h * a.w;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < a.c; ++c)
    for (size_t i = 0; i < hw; ++i) y.d[c * hw + i] = a.d[c * hw + i] + b.d[c * hw + i];
  keep(n->r, name, y, 0);
  return y;
}

static ten cat(net *n, const char *name, const ten *ts, int cnt) {
  int c = 0;
  for (int i = 0; i < cnt; ++i) c += ts[i].c;
  ten y = new_ten(n->r, c, ts[0].h, ts[0].w);
  if (n->r->failed) return y;
  size_t off = 0;
  const size_t hw = (size_t)ts[0].h * ts[0].w;
  for (int i = 0; i < cnt; ++i) {
    float *dst = y.d + off;
    const float *srcp = ts[i].d;
#pragma omp parallel for schedule(static)
    for (int ch = 0; ch < ts[i].c; ++ch) memcpy(dst + ch * hw, srcp + ch * hw, hw * sizeof(float));
    off += (size_t)ts[i].c * hw;
  }
  keep(n->r, name, y, 0);
  return y;
}

/* Bottleneck inside C3k2: expansion=1.0, shortcut=True (model.py:71-73, 99): x + cv2(cv1(x)) */
static ten bottleneck(net *n, const char *name, ten x) {
  char s[256];
  snprintf(s, sizeof s, "%s.cv1", name);
  ten t = conv_block(n, s, x, x.c, 1, 1);
  if (n->r->failed) return t;
  snprintf(s, sizeof s, "%s.cv2", name);
  t = conv_block(n, s, t, x.c, 3, 1);
  if (n->r->failed) return t;
  snprintf(s, sizeof s, "%s.add", name);
  return add(n, s, x, t);
}

/* C3k2 (model.py:76-110): cv3(cat[bottlenecks(cv1(x)), cv2(x)]) */
static ten c3k2(net *n, const char *name, ten x, int cout, int nb) {
  char s[256];
  const int hid = cout / 2;
  snprintf(s, sizeof s, "%s.cv1", name);
  ten p1 = conv_block(n, s, x, hid, 1, 1);
  if (n->r->failed) return p1;
  snprintf(s, sizeof s, "%s.cv2", name);
  ten p2 = conv_block(n, s, x, hid, 1, 1);
  if (n->r->failed) return p2;
  for (int i = 0; i < nb; ++i) {
    snprintf(s, sizeof s, "%s.bottlenecks.%d", name, i);
    p1 = bottleneck(n, s, p1);
    if (n->r->failed) return p1;
  }
  ten pair[2] = {p1, p2};
  snprintf(s, sizeof s, "%s.cat", name);
  ten c = cat(n, s, pair, 2);
  if (n->r->failed) return c;
  snprintf(s, sizeof s, "%s.cv3", name);
  return conv_block(n, s, c, cout, 1, 1);
}

/* MaxPool2d(5,1,2): window clipped at the border == -inf padding (model.py:125) */
static ten maxpool5(net *n, const char *name, ten x) {
  ten y = new_ten(n->r, x.c, x.h, x.w);
  if (n->r->failed) return y;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < x.c; ++c)
    for (int i = 0; i < x.h; ++i)
      for (int j = 0; j < x.w; ++j) {
        float m = -INFINITY;
        for (int di = -2; di <= 2; ++di) {
          int ii = i + di;
          if (ii < 0 || ii >= x.h) continue;
          for (int dj = -2; dj <= 2; ++dj) {
            int jj = j + dj;
            if (jj < 0 || jj >= x.w) continue;
            float v = x.d[((size_t)c * x.h + ii) * x.w + jj];
            if (v > m) m = v;
          }
        }
        y.d[((size_t)c * x.h + i) * x.w + j] = m;
      }
  keep(n->r, name, y, 0);
  return y;
}

/* SPPF_DLA (model.py:113-132) */
static ten sppf(net *n, const char *name, ten x, int cout) {
  char s[256];
  snprintf(s, sizeof s, "%s.cv1", name);
  ten t[4];
  t[0] = conv_block(n, s, x, x.c / 2, 1, 1);
  if (n->r->failed) return t[0];
  for (int i = 1; i < 4; ++i) {
    snprintf(s, sizeof s, "%s.pool%d", name, i);
    t[i] = maxpool5(n, s, t[i - 1]);
    if (n->r->failed) return t[i];
  }
  snprintf(s, sizeof s, "%s.cat", name);
  ten c = cat(n, s, t, 4);
  if (n->r->failed) return c;
  snprintf(s, sizeof s, "%s.cv2", name);
  return conv_block(n, s, c, cout, 1, 1);
}

/* Upsample nearest x2 (model.py:145-147): out[y][x] = in[y/2][x/2] */
static ten up2(net *n, const char *name, ten x) {
  ten y = new_ten(n->r, x.c, 2 * x.h, 2 * x.w);
  if (n->r->failed) return y;
#pragma omp parallel for schedule(static)
  for (int c = 0; c < x.c; ++c)
    for (int i = 0; i < y.h; ++i)
      for (int j = 0; j < y.w; ++j)
        y.d[((size_t)c * y.h + i) * y.w + j] = x.d[((size_t)c * x.h + i / 2) * x.w + j / 2];
  keep(n->r, name, y, 0);
  return y;
}

/* DetectionHead branch (model.py:289-303): ConvBlock3x3 -> ConvBlock3x3 -> Conv2d 1x1 (+bias) */
static ten head_branch(net *n, const char *head, const char *branch, ten x, int nout, const char *out_name) {
  char s[256];
  snprintf(s, sizeof s, "%s.%s.0", head, branch);
  ten t = conv_block(n, s, x, x.c, 3, 1);
  if (n->r->failed) return t;
  snprintf(s, sizeof s, "%s.%s.1", head, branch);
  t = conv_block(n, s, t, x.c, 3, 1);
  if (n->r->failed) return t;
  snprintf(s, sizeof s, "%s.%s.2", head, branch);
  const float *w = param(n, s, "weight", nout * x.c), *b = param(n, s, "bias", nout);
  if (n->r->failed) return t;
  ten y = conv2d(n->r, t, w, b, nout, 1, 1);
  keep(n->r, out_name, y, 1);
  return y;
}

uo_run *uo_forward(const uo_statedict *sd, const float *x, int H, int W, int num_classes, int base_channels,
                   int lite_p2, int keep_all, int nthreads) {
  if (g_slab_live) {
    set_err("uo_forward: previous run not freed (one live run at a time)");
    return NULL;
  }
  if (H % 16 || W % 16) {
    set_err("H and W must be multiples of 16");
    return NULL;
  }
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  /* activation volume: ~263 floats per input pixel for base_channels=32; be generous */
  size_t need = (size_t)H * W * 420 * (size_t)(base_channels > 32 ? base_channels / 32 : 1) * sizeof(float) + (64u << 20);
  if (need > g_slab_cap) {
    free(g_slab);
    g_slab = NULL;
    if (posix_memalign((void **)&g_slab, 64, need)) {
      g_slab_cap = 0;
      set_err("oracle: cannot allocate %zu bytes", need);
      return NULL;
    }
    memset(g_slab, 0, need);
    g_slab_cap = need;
  }
  g_slab_used = 0;
  g_slab_live = 1;

  uo_run *r = calloc(1, sizeof *r);
  r->keep_all = keep_all;
  net n = {r, sd};
  const int c1 = base_channels, c2 = c1 * 2, c3 = c1 * 4, c4 = c1 * 8;

  ten in = new_ten(r, 3, H, W);
  if (r->failed) return r;
  memcpy(in.d, x, sizeof(float) * 3 * (size_t)H * W);

  /* Backbone.forward (model.py:205-219) */
  ten t = conv_block(&n, "backbone.stem", in, c1, 3, 2);
  if (r->failed) return r;
  t = conv_block(&n, "backbone.stage1_conv", t, c2, 3, 2);
  if (r->failed) return r;
  ten p2 = lite_p2 ? conv_block(&n, "backbone.stage1_block", t, c2, 3, 1) : c3k2(&n, "backbone.stage1_block", t, c2, 1);
  if (r->failed) return r;
  t = conv_block(&n, "backbone.stage2_conv", p2, c3, 3, 2);
  if (r->failed) return r;
  ten p3 = c3k2(&n, "backbone.stage2_c3k2", t, c3, 2);
  if (r->failed) return r;
  t = conv_block(&n, "backbone.stage3_conv", p3, c4, 3, 2);
  if (r->failed) return r;
  ten p4 = c3k2(&n, "backbone.stage3_c3k2", t, c4, 2);
  if (r->failed) return r;
  ten p4s = sppf(&n, "backbone.sppf", p4, c4);
  if (r->failed) return r;

  /* Neck.forward (model.py:252-269) */
  t = conv_block(&n, "neck.lateral_p3", p4s, c3, 1, 1);
  if (r->failed) return r;
  ten pr[2];
  pr[0] = up2(&n, "neck.up1", t);
  pr[1] = p3;
  if (r->failed) return r;
  t = cat(&n, "neck.cat_fpn1", pr, 2);
  if (r->failed) return r;
  ten p3f = c3k2(&n, "neck.fpn_c3k2_1", t, c3, 1);
  if (r->failed) return r;
  t = conv_block(&n, "neck.lateral_p2", p3f, c2, 1, 1);
  if (r->failed) return r;
  pr[0] = up2(&n, "neck.up2", t);
  pr[1] = p2;
  if (r->failed) return r;
  t = cat(&n, "neck.cat_fpn2", pr, 2);
  if (r->failed) return r;
  ten p2f = c3k2(&n, "neck.fpn_c3k2_2", t, c2, 1);
  if (r->failed) return r;
  pr[0] = conv_block(&n, "neck.down1", p2f, c2, 3, 2);
  pr[1] = p3f;
  if (r->failed) return r;
  t = cat(&n, "neck.cat_pan1", pr, 2);
  if (r->failed) return r;
  ten p3o = c3k2(&n, "neck.pan_c3k2_1", t, c3, 1);
  if (r->failed) return r;
  pr[0] = conv_block(&n, "neck.down2", p3o, c3, 3, 2);
  pr[1] = p4; /* PRE-SPPF p4 (model.py:254,267) */
  if (r->failed) return r;
  t = cat(&n, "neck.cat_pan2", pr, 2);
  if (r->failed) return r;
  ten p4o = c3k2(&n, "neck.pan_c3k2_2", t, c4, 1);
  if (r->failed) return r;

  /* heads (model.py:361-365) */
  head_branch(&n, "head_p2", "cls_branch", p2f, num_classes, "p2_cls");
  if (r->failed) return r;
  head_branch(&n, "head_p2", "reg_branch", p2f, 4, "p2_reg");
  if (r->failed) return r;
  head_branch(&n, "head_p3", "cls_branch", p3o, num_classes, "p3_cls");
  if (r->failed) return r;
  head_branch(&n, "head_p3", "reg_branch", p3o, 4, "p3_reg");
  if (r->failed) return r;
  head_branch(&n, "head_p4", "cls_branch", p4o, num_classes, "p4_cls");
  if (r->failed) return r;
  head_branch(&n, "head_p4", "reg_branch", p4o, 4, "p4_reg");
  return r;
}

/* DetectionHead branch of the QAT model: head_pN_cls / head_pN_reg are nn.Sequential(QuantConvBlock 3x3,
 * QuantConvBlock 3x3, nn.Conv2d 1x1)  (qat.py:411-440) -- same arithmetic as model.py:289-303, flat names. */
static ten head_seq(net *n, const char *seq, ten x, int nout, const char *out_name) {
  char s[256];
  snprintf(s, sizeof s, "%s.0", seq);
  ten t = conv_block(n, s, x, x.c, 3, 1);
  if (n->r->failed) return t;
  snprintf(s, sizeof s, "%s.1", seq);
  t = conv_block(n, s, t, x.c, 3, 1);
  if (n->r->failed) return t;
  snprintf(s, sizeof s, "%s.2", seq);
  const float *w = param(n, s, "weight", nout * x.c), *b = param(n, s, "bias", nout);
  if (n->r->failed) return t;
  ten y = conv2d(n->r, t, w, b, nout, 1, 1);
  keep(n->r, out_name, y, 1);
  return y;
}

/* UNINA_YOLO_DLA_QAT.forward (qat.py:443-491) in float: the model the reference's QAT checkpoints belong to
 * (graph (B): stride-32 stage + SPPF at 16x base channels, three FPN levels, PAN concat on the FUSED p4).
 * Without pytorch-quantization the reference itself runs exactly this (QuantConvBlock falls back to nn.Conv2d,
 * qat.py:249-254; QuantBottleneck adds x unquantised, qat.py:290-292). */
uo_run *uo_forward_qat(const uo_statedict *sd, const float *x, int H, int W, int num_classes, int base_channels,
                       int keep_all, int nthreads) {
  if (g_slab_live) {
    set_err("uo_forward_qat: previous run not freed (one live run at a time)");
    return NULL;
  }
  if (H % 32 || W % 32) {
    set_err("H and W must be multiples of 32");
    return NULL;
  }
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
  size_t need = (size_t)H * W * 480 * (size_t)(base_channels > 32 ? base_channels / 32 : 1) * sizeof(float) + (64u << 20);
  if (need > g_slab_cap) {
    free(g_slab);
    g_slab = NULL;
    if (posix_memalign((void **)&g_slab, 64, need)) {
      g_slab_cap = 0;
      set_err("oracle: cannot allocate %zu bytes", need);
      return NULL;
    }
    memset(g_slab, 0, need);
    g_slab_cap = need;
  }
  g_slab_used = 0;
  g_slab_live = 1;

  uo_run *r = calloc(1, sizeof *r);
  r->keep_all = keep_all;
  net n = {r, sd};
  const int c1 = base_channels, c2 = c1 * 2, c3 = c1 * 4, c4 = c1 * 8, c5 = c1 * 16;

  ten in = new_ten(r, 3, H, W);
  if (r->failed) return r;
  memcpy(in.d, x, sizeof(float) * 3 * (size_t)H * W);

#define CHK if (r->failed) return r
  /* backbone (qat.py:445-458) */
  ten t = conv_block(&n, "stem", in, c1, 3, 2); CHK;
  t = conv_block(&n, "stage1_conv", t, c2, 3, 2); CHK;
  ten p2 = c3k2(&n, "stage1_c3k2", t, c2, 1); CHK;
  t = conv_block(&n, "stage2_conv", p2, c3, 3, 2); CHK;
  ten p3 = c3k2(&n, "stage2_c3k2", t, c3, 2); CHK;
  t = conv_block(&n, "stage3_conv", p3, c4, 3, 2); CHK;
  ten p4 = c3k2(&n, "stage3_c3k2", t, c4, 2); CHK;
  t = conv_block(&n, "stage4_conv", p4, c5, 3, 2); CHK;
  ten p5 = sppf(&n, "stage4_sppf", t, c5); CHK;

  /* neck, top-down (qat.py:461-470) */
  ten pr[2];
  t = conv_block(&n, "lateral_p4", p5, c4, 1, 1); CHK;
  pr[0] = up2(&n, "up_p5", t); pr[1] = p4; CHK;
  t = cat(&n, "cat_fpn1", pr, 2); CHK;
  ten p4f = c3k2(&n, "fpn_c3k2_1", t, c4, 1); CHK;
  t = conv_block(&n, "lateral_p3", p4f, c3, 1, 1); CHK;
  pr[0] = up2(&n, "up_p4", t); pr[1] = p3; CHK;
  t = cat(&n, "cat_fpn2", pr, 2); CHK;
  ten p3f = c3k2(&n, "fpn_c3k2_2", t, c3, 1); CHK;
  t = conv_block(&n, "lateral_p2", p3f, c2, 1, 1); CHK;
  pr[0] = up2(&n, "up_p3", t); pr[1] = p2; CHK;
  t = cat(&n, "cat_fpn3", pr, 2); CHK;
  ten p2f = c3k2(&n, "fpn_c3k2_3", t, c2, 1); CHK;
  /* bottom-up (qat.py:473-477) */
  pr[0] = conv_block(&n, "down1", p2f, c2, 3, 2); pr[1] = p3f; CHK;
  t = cat(&n, "cat_pan1", pr, 2); CHK;
  ten p3o = c3k2(&n, "pan_c3k2_1", t, c3, 1); CHK;
  pr[0] = conv_block(&n, "down2", p3o, c3, 3, 2); pr[1] = p4f; /* the FUSED p4 (qat.py:476) */ CHK;
  t = cat(&n, "cat_pan2", pr, 2); CHK;
  ten p4o = c3k2(&n, "pan_c3k2_2", t, c4, 1); CHK;

  /* heads (qat.py:480-489) */
  head_seq(&n, "head_p2_cls", p2f, num_classes, "p2_cls"); CHK;
  head_seq(&n, "head_p2_reg", p2f, 4, "p2_reg"); CHK;
  head_seq(&n, "head_p3_cls", p3o, num_classes, "p3_cls"); CHK;
  head_seq(&n, "head_p3_reg", p3o, 4, "p3_reg"); CHK;
  head_seq(&n, "head_p4_cls", p4o, num_classes, "p4_cls"); CHK;
  head_seq(&n, "head_p4_reg", p4o, 4, "p4_reg");
#undef CHK
  return r;
}

const float *uo_run_get(const uo_run *r, const char *name, int *c, int *h, int *w) {
  if (!r || r->failed) return NULL;
  for (int i = 0; i < r->n; ++i)
    if (!strcmp(r->names[i], name)) {
      if (c) *c = r->t[i].c;
      if (h) *h = r->t[i].h;
      if (w) *w = r->t[i].w;
      return r->t[i].d;
    }
  return NULL;
}
int uo_run_count(const uo_run *r) { return r && !r->failed ? r->n : -1; }
const char *uo_run_name(const uo_run *r, int i) { return r->names[i]; }
void uo_run_free(uo_run *r) {
  if (!r) return;
  for (int i = 0; i < r->n; ++i) free(r->names[i]);
  free(r);
  g_slab_live = 0;
  g_slab_used = 0;
}
