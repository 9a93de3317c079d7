// train_ops.hip — the small kernels of the TRAINING graph: weight-norm forward /
// backward, transposed weight packing for conv dgrad, bias / broadcast reductions, element-wise forward / backward
// ops, decimation (strided and transposed convolutions are lowered to dense MFMA convolutions over phase-decimated
// signals), grouped strided Conv1d (DiscriminatorS, models.py:206-211), scalar loss reductions and the fused
// AdamW step (train.py:79-88,191-213).  All HBM-bound.
#include "common.h"

namespace {

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  const int tid = threadIdx.x;
  sh[tid] = v;
  __syncthreads();
  for (int s = blockDim.x >> 1; s > 0; s >>= 1) {
    if (tid < s) sh[tid] += sh[tid + s];
    __syncthreads();
  }
  const double r = sh[0];
  __syncthreads();
  return r;
}

// ---- weight norm (torch.nn.utils.weight_norm, dim=0): w[r,:] = g[r] * v[r,:] / ||v[r,:]|| ---------------------
__global__ void weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ g, float* __restrict__ w,
                                       float* __restrict__ norm, int cols) {
  __shared__ double sh[256];
  const int r = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < cols; i += blockDim.x) {
    const double x = v[(long long)r * cols + i];
    acc += x * x;
  }
  const float nr = (float)sqrt(block_sum_d(acc, sh));
  const float sc = g[r] / nr;
  if (threadIdx.x == 0 && norm) norm[r] = nr;
  for (int i = threadIdx.x; i < cols; i += blockDim.x) w[(long long)r * cols + i] = v[(long long)r * cols + i] * sc;
}

// dg[r] = <dw, v> / ||v|| ;  dv = (g/||v||) * (dw - v * <dw,v> / ||v||^2)
__global__ void weight_norm_bwd_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                       const float* __restrict__ norm, const float* __restrict__ dw,
                                       float* __restrict__ dv, float* __restrict__ dg, int cols) {
  __shared__ double sh[256];
  const int r = blockIdx.x;
  double acc = 0.0;
  for (int i = threadIdx.x; i < cols; i += blockDim.x)
    acc += (double)dw[(long long)r * cols + i] * (double)v[(long long)r * cols + i];
  const double dot = block_sum_d(acc, sh);
  const double nr = norm[r];
  if (threadIdx.x == 0) dg[r] = (float)(dot / nr);
  const double sc = (double)g[r] / nr;
  const double k = dot / (nr * nr);
  for (int i = threadIdx.x; i < cols; i += blockDim.x) {
    const long long o = (long long)r * cols + i;
    dv[o] = (float)(sc * ((double)dw[o] - (double)v[o] * k));
  }
}

// dgrad weight: dst[co][KS-1-k][ci] = w[co][ci][k]  (CinP-padded rows of zeros) — the conv that maps dy -> dx
__global__ void pack_conv1d_T_kernel(const float* __restrict__ w, float* __restrict__ dst, int Cout, int Cin, int KS,
                                     int CinP) {
  const long long n = (long long)Cout * KS * CinP;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % CinP);
    const long long r = i / CinP;
    const int kk = (int)(r % KS);
    const int co = (int)(r / KS);
    dst[i] = ci < Cin ? w[((long long)co * Cin + ci) * KS + (KS - 1 - kk)] : 0.f;
  }
}

// ---- one-launch weight preparation of a training convolution --------------------------------------------------------
// Every convolution of the training graph is lowered to a dense stride-1 MFMA convolution whose weight wd[o][i][m] is an
// index map of the parameter v[r][c][k] (times the weight-norm scale g[r]/||v[r]||):
//   kind 0 dense       wd[r][c][k]                        = w[r][c][k]
//   kind 1 strided(s)  wd[r][q*C2 + c][m], s*m + q = k+shift  (svc_decimate_f32 turns the s phases into channels)
//   kind 2 transposed  wd[ph*C2 + c][r][M-1-mm], k = ph + mm*u  (v is [Cin][Cout][K]; the u output phases become channels)
// This kernel writes the forward operand wp[(i*Kd + m)*OdP + o] (svc_pack_conv1d_weight's layout) AND the dgrad operand
// wt[(o*Kd + Kd-1-m)*IdP + i] (svc_pack_conv1d_weight_T's) straight from v: one launch instead of weight_norm_fwd +
// (pad, permute, copy) + pack in the forward and pack_T in the backward.  Entries no (r,c,k) maps to — the OdP / IdP padding
// and the taps past K of a strided / transposed layout — are never written: the caller allocates wp / wt zero-filled ONCE.
struct WMap {
  int kind, R, C2, K, Od, Id, Kd, OdP, IdP, s, shift;
};
__device__ __forceinline__ void wmap_index(const WMap& p, int r, int c, int k, int& o, int& i, int& m) {
  if (p.kind == 0) {
    o = r; i = c; m = k;
  } else if (p.kind == 1) {
    const int kk = k + p.shift;
    m = kk / p.s;
    o = r;
    i = (kk - m * p.s) * p.C2 + c;
  } else {
    const int mm = k / p.s, ph = k - mm * p.s;
    o = ph * p.C2 + c;
    i = r;
    m = p.Kd - 1 - mm;
  }
}
// Two launches: row norms (one workgroup per weight-normed row), then the scatter.  A scatter workgroup owns WP_ROWS parameter
// rows x a range of input-channel columns (~WP_COLS elements per row) and writes both operand arrays with the lanes of a half
// wave along the array's FASTEST index, so that every store instruction fills whole 128-byte lines:
//   wp [i][m][o]: o = r for dense / strided layers (lanes along rows), o = ph*C2 + c for transposed ones (lanes along columns);
//   wt [o][m][i]: i = q*C2 + c for dense / strided layers (lanes along columns), i = r for transposed ones (lanes along rows).
// (Round 2 ran one workgroup per row and stored element j of the row at stride OdP / IdP floats: 4-byte stores to a different
// line each — the 1024 x 1024 x 5 period convs made the one-launch form spend 640 us per launch, profiles/r04q_train_*; whole
// rows per workgroup instead left those same weights with 32 workgroups of 160 k elements each.)
// The reads of the "lanes along rows" pass touch 32 lines at a time (one per row), each consumed within two trips: L1 hits.
constexpr int WP_ROWS = SVC_WEIGHT_PREP_ROWS, WP_COLS = SVC_WEIGHT_PREP_COLS;
static_assert(WP_ROWS == 32, "the row pass maps one lane of a half wave to one row");
__host__ __device__ inline int wp_cols_per_block(int K) { return max(1, WP_COLS / K); }      // input channels per workgroup
__host__ __device__ inline int wp_blocks(int R, int C2, int K) {
  const int cpc = wp_cols_per_block(K);
  return ((R + WP_ROWS - 1) / WP_ROWS) * ((C2 + cpc - 1) / cpc);
}
__device__ __forceinline__ void conv_weight_norm_row(const float* __restrict__ v, float* __restrict__ norm, int n, int r, double* sh) {
  const float* vr = v + (long long)r * n;
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) {
    const double x = vr[j];
    acc += x * x;
  }
  const float nr = (float)sqrt(block_sum_d(acc, sh));
  if (threadIdx.x == 0) norm[r] = nr;
}
__device__ __forceinline__ void conv_weight_scatter_block(const float* __restrict__ v, const float* __restrict__ g,
                                                          float* __restrict__ wp, float* __restrict__ wt,
                                                          const float* __restrict__ norm, const WMap& p, int blk) {
  const int n = p.C2 * p.K;
  const int cpc = wp_cols_per_block(p.K), n_chunks = (p.C2 + cpc - 1) / cpc;
  const int grp = blk / n_chunks, chunk = blk - grp * n_chunks;
  const int r0 = grp * WP_ROWS, nrow = min(WP_ROWS, p.R - r0);
  const int c_lo = chunk * cpc, c_hi = min(c_lo + cpc, p.C2);
  const int tid = threadIdx.x, rl = tid & 31, jj = tid >> 5;          // 8 column walkers per row
  const bool transposed = p.kind == 2;
  // ---- lanes along rows: wp for dense / strided, wt for transposed
  if (rl < nrow) {
    const int r = r0 + rl;
    const float* vr = v + (long long)r * n;
    const float s = g ? g[r] / norm[r] : 1.f;
    for (int j = c_lo * p.K + jj; j < c_hi * p.K; j += 8) {
      const int c = j / p.K, k = j - c * p.K;
      int o, i, m;
      wmap_index(p, r, c, k, o, i, m);
      const float val = vr[j] * s;
      if (!transposed) wp[((long long)i * p.Kd + m) * p.OdP + o] = val;
      else if (wt) wt[((long long)o * p.Kd + (p.Kd - 1 - m)) * p.IdP + i] = val;
    }
  }
  // ---- lanes along columns: wt for dense / strided, wp for transposed (a wave per row, tap by tap)
  const int wave = tid >> 6, lane = tid & 63;
  for (int rr = wave; rr < nrow; rr += 4) {
    const int r = r0 + rr;
    const float* vr = v + (long long)r * n;
    const float s = g ? g[r] / norm[r] : 1.f;
    for (int k = 0; k < p.K; ++k) {
      for (int c = c_lo + lane; c < c_hi; c += 64) {
        int o, i, m;
        wmap_index(p, r, c, k, o, i, m);
        const float val = vr[c * p.K + k] * s;
        if (transposed) wp[((long long)i * p.Kd + m) * p.OdP + o] = val;
        else if (wt) wt[((long long)o * p.Kd + (p.Kd - 1 - m)) * p.IdP + i] = val;
      }
    }
  }
}
__device__ __forceinline__ WMap wmap_of(const svc_conv_weight_args& a) {
  WMap p;
  p.kind = a.kind; p.R = a.R; p.C2 = a.C2; p.K = a.K; p.Od = a.Od; p.Id = a.Id; p.Kd = a.Kd; p.OdP = a.OdP; p.IdP = a.IdP;
  p.s = a.s; p.shift = a.shift;
  return p;
}
__device__ __forceinline__ int plan_of(const int* __restrict__ start, int n_plans, int x) {   // last plan whose start is <= x
  int lo = 0, hi = n_plans - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (start[mid] <= x) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void conv_weight_norm_kernel(const float* __restrict__ v, float* __restrict__ norm, int n) {
  __shared__ double sh[256];
  conv_weight_norm_row(v, norm, n, blockIdx.x, sh);
}
__global__ __launch_bounds__(256) void conv_weight_scatter_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                                                  float* __restrict__ wp, float* __restrict__ wt,
                                                                  const float* __restrict__ norm, WMap p) {
  conv_weight_scatter_block(v, g, wp, wt, norm, p, blockIdx.x);
}
// the same for MANY convolutions: workgroup -> (plan, row) resp. (plan, block) through prefix sums over the plans.  A training
// iteration prepares ~360 weights (generator + discriminators, the latter twice), each a 5..12 us launch that is latency- not
// bandwidth-bound (profiles/r03l_train_*: 359 x 11.8 us = 4.2 ms of a 110 ms step); together they move ~1.5 GB.
__global__ __launch_bounds__(256) void conv_weight_norm_multi_kernel(const svc_conv_weight_args* __restrict__ tab,
                                                                     const int* __restrict__ row_start, int n_plans) {
  __shared__ double sh[256];
  const int lo = plan_of(row_start, n_plans, blockIdx.x);
  const svc_conv_weight_args a = tab[lo];
  if (a.g) conv_weight_norm_row(a.v, a.norm, a.C2 * a.K, blockIdx.x - row_start[lo], sh);   // (plain weights: nothing to do)
}
__global__ __launch_bounds__(256) void conv_weight_scatter_multi_kernel(const svc_conv_weight_args* __restrict__ tab,
                                                                        const int* __restrict__ block_start, int n_plans) {
  const int lo = plan_of(block_start, n_plans, blockIdx.x);
  const svc_conv_weight_args a = tab[lo];
  const WMap p = wmap_of(a);
  conv_weight_scatter_block(a.v, a.g, a.wp, a.wt, a.norm, p, blockIdx.x - block_start[lo]);
}
// adjoint: dwd [Od][Id][Kd] (svc_conv1d_wgrad_f32's output) -> dv (and dg): dw[r][c][k] = dwd[o][i][m],
// dg[r] = <dw, v> / ||v||,  dv = (g/||v||) (dw - v <dw,v>/||v||^2);  without g: dv = dw.
__global__ void conv_weight_grad_kernel(const float* __restrict__ v, const float* __restrict__ g,
                                        const float* __restrict__ norm, const float* __restrict__ dwd,
                                        float* __restrict__ dv, float* __restrict__ dg, WMap p) {
  __shared__ double sh[256];
  const int r = blockIdx.x;
  const int n = p.C2 * p.K;
  const float* vr = v + (long long)r * n;
  auto dw_at = [&](int j) {
    const int c = j / p.K, k = j - c * p.K;
    int o, i, m;
    wmap_index(p, r, c, k, o, i, m);
    return dwd[((long long)o * p.Id + i) * p.Kd + m];
  };
  if (!g) {
    for (int j = threadIdx.x; j < n; j += blockDim.x) dv[(long long)r * n + j] = dw_at(j);
    return;
  }
  double acc = 0.0;
  for (int j = threadIdx.x; j < n; j += blockDim.x) acc += (double)dw_at(j) * (double)vr[j];
  const double dot = block_sum_d(acc, sh);
  const double nr = norm[r];
  if (threadIdx.x == 0) dg[r] = (float)(dot / nr);
  const double sc = (double)g[r] / nr;
  const double kq = dot / (nr * nr);
  for (int j = threadIdx.x; j < n; j += blockDim.x)
    dv[(long long)r * n + j] = (float)(sc * ((double)dw_at(j) - (double)vr[j] * kq));
}

// ---- reductions of [B,C,T] -----------------------------------------------------------------------------------
// mode 0: out[c] = sum_{b,t}   (bias grads)     one block per c
// mode 1: out[b,c] = sum_t     (grad of a [B,C,1] broadcast)   one block per (b,c)
__global__ void reduce_bct_kernel(const float* __restrict__ x, float* __restrict__ out, long long bs, long long cs,
                                  int B, int C, int T, int mode, float beta) {
  __shared__ double sh[256];
  double acc = 0.0;
  if (mode == 0) {
    const int c = blockIdx.x;
    for (int b = 0; b < B; ++b)
      for (int t = threadIdx.x; t < T; t += blockDim.x) acc += x[b * bs + c * cs + t];
  } else {
    const int b = blockIdx.x / C, c = blockIdx.x % C;
    for (int t = threadIdx.x; t < T; t += blockDim.x) acc += x[b * bs + c * cs + t];
  }
  const double s = block_sum_d(acc, sh);
  if (threadIdx.x == 0) out[blockIdx.x] = (float)s + (beta != 0.f ? beta * out[blockIdx.x] : 0.f);
}

// out[b,t] = sum_c x[b,c,t] * (w ? w[c] : 1)     (mode 2)
__global__ void reduce_c_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out,
                                int C, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= T) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) acc += x[((long long)b * C + c) * T + t] * (w ? w[c] : 1.f);
  out[(long long)b * T + t] = acc;
}

// ---- element-wise --------------------------------------------------------------------------------------------
__device__ __forceinline__ float ew_apply(int op, float a, float b, float alpha, float beta) {
  switch (op) {
    case SVC_EW_ADD: return alpha * a + beta * b;
    case SVC_EW_MUL: return a * b * alpha;
    case SVC_EW_LRELU: return a > 0.f ? a : a * alpha;
    case SVC_EW_LRELU_BWD: return b > 0.f ? a : a * alpha;            // a = dy, b = x
    case SVC_EW_TANH: return tanhf(a);
    case SVC_EW_TANH_BWD: return a * (1.f - b * b);                   // a = dy, b = y
    case SVC_EW_RELU: return a > 0.f ? a : 0.f;
    case SVC_EW_RELU_BWD: return b > 0.f ? a : 0.f;                   // a = dy, b = x
    case SVC_EW_EXP: return expf(a * alpha);
    case SVC_EW_LOG_CLAMP: return logf(fmaxf(a, alpha));              // log(clamp(a, min=alpha))
    case SVC_EW_LOG_CLAMP_BWD: return b > alpha ? a / b : 0.f;        // a = dy, b = x
    case SVC_EW_SCALE: return a * alpha + beta;
    case SVC_EW_SIGMOID: return svc_sigmoid(a);
    case SVC_EW_SQUARE: return a * a * alpha;
    case SVC_EW_SIGN_MUL: return (a > 0.f ? 1.f : (a < 0.f ? -1.f : 0.f)) * alpha;   // d|a|/da * alpha
    case SVC_EW_DIV: return a / b * alpha;
    case SVC_EW_GELU: return svc_gelu(a);
    case SVC_EW_MISH: return a * tanhf(a > 20.f ? a : log1pf(expf(a)));   // softplus with torch's threshold 20
    case SVC_EW_CLAMP: return fminf(fmaxf(a, alpha), beta);
    case SVC_EW_MISH_BWD: {                                           // a = dy, b = x
      const float sp = b > 20.f ? b : log1pf(expf(b));
      const float th = tanhf(sp);
      return a * (th + b * (1.f - th * th) * svc_sigmoid(b));
    }
    case SVC_EW_DROPOUT: return b >= alpha ? a * (1.f / (1.f - alpha)) : 0.f;   // b = uniform draw, alpha = p
    default: return a;
  }
}

__global__ void lrelu_bwd_add_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ r,
                                     float* __restrict__ y, long long n, float slope) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = (x[i] > 0.f ? dy[i] : dy[i] * slope) + r[i];
}

__global__ void ew_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                          long long n, float alpha, float beta) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    y[i] = ew_apply(op, a[i], b ? b[i] : 0.f, alpha, beta);
}

// y[b,c,t] = op(x[b,c,t], side[b*s_bs + c*s_cs + t*s_ts])  with arbitrary (possibly negative / zero) strides
__global__ void ew_bct_kernel(int op, const float* __restrict__ x, const float* __restrict__ side, float* __restrict__ y,
                              long long x_bs, long long x_cs, long long s_bs, long long s_cs, long long s_ts,
                              long long y_bs, long long y_cs, int C, int T, float alpha, float beta) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float sv = side[b * s_bs + c * s_cs + t * s_ts];
  y[b * y_bs + c * y_cs + t] = ew_apply(op, x[b * x_bs + c * x_cs + t], sv, alpha, beta);
}

// WN gate: in [B,2H,T] -> acts[B,H,T] = tanh(in[:, :H]) * sigmoid(in[:, H:])   (modules/commons.py:129-136)
__global__ void gate_fwd_kernel(const float* __restrict__ in, float* __restrict__ acts, int H, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const float a = in[((long long)b * 2 * H + c) * T + t], s = in[((long long)b * 2 * H + H + c) * T + t];
  acts[((long long)b * H + c) * T + t] = tanhf(a) * svc_sigmoid(s);
}
__global__ void gate_bwd_kernel(const float* __restrict__ in, const float* __restrict__ dacts, float* __restrict__ din,
                                int H, int T) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  const long long oa = ((long long)b * 2 * H + c) * T + t, os = ((long long)b * 2 * H + H + c) * T + t;
  const float th = tanhf(in[oa]), sg = svc_sigmoid(in[os]);
  const float d = dacts[((long long)b * H + c) * T + t];
  din[oa] = d * sg * (1.f - th * th);
  din[os] = d * th * sg * (1.f - sg);
}

// ---- decimation: y[b, r*C + c, q*w + j] = xpad[b, c, (q*s + r + off)*w + j],  j < w ------------------------------
// The signal is a sequence of blocks of `w` samples (w = 1: plain samples; w = period p: the rows of DiscriminatorP's
// [T/p, p] view kept time-contiguous) and the decimation runs over the BLOCK index.  xpad = x reflect-padded on the
// right up to `lp` samples when lp > T (DiscriminatorP, models.py:185-189), zero elsewhere.  Lowers a stride-s conv
// (and the dgrad/wgrad of a transposed conv) to a dense conv (dilation w) over s*C channels.
__global__ void decimate_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int T, int s, int w, int off,
                                int Q, int lp) {
  const int qq = blockIdx.x * blockDim.x + threadIdx.x;
  const int rc = blockIdx.y, b = blockIdx.z;
  if (qq >= Q * w) return;
  const int r = rc / C, c = rc % C;
  const int q = qq / w, j = qq - q * w;
  const int blk = q * s + r + off;
  int tau = blk * w + j;
  float v = 0.f;
  if (blk >= 0 && tau < lp) {
    if (tau >= T) tau = 2 * T - 2 - tau;
    if (tau >= 0) v = x[((long long)b * C + c) * T + tau];
  }
  y[((long long)b * s * C + rc) * ((long long)Q * w) + qq] = v;
}
// adjoint: dx[b,c,tau] = sum of dy over every (r,q) that read tau (direct + reflected image)
__global__ void decimate_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int C, int T, int s, int w,
                                    int off, int Q, int lp) {
  const int tau = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y, b = blockIdx.z;
  if (tau >= T) return;
  float acc = 0.f;
  auto take = [&](int tp) {   // padded position tp
    if (tp < 0 || tp >= lp) return;
    const int blk = tp / w, j = tp - blk * w;
    const int u = blk - off;
    if (u < 0) return;
    const int q = u / s, r = u - q * s;
    if (q < Q) acc += dy[((long long)b * s * C + r * C + c) * ((long long)Q * w) + q * w + j];
  };
  take(tau);
  const int img = 2 * T - 2 - tau;
  if (img >= T && img < lp) take(img);
  dx[((long long)b * C + c) * T + tau] = acc;
}

// ---- grouped strided Conv1d (DiscriminatorS: k=41, stride 4, groups 4..256; models.py:206-211) ------------------
// w [Cout][Cg][KS], Cg = Cin/groups, Og = Cout/groups
__global__ void gconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                 float* __restrict__ y, int Cin, int Cout, int Tin, int Tout, int KS, int stride, int pad,
                                 int groups) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int co = blockIdx.y, b = blockIdx.z;
  if (t >= Tout) return;
  const int Cg = Cin / groups, Og = Cout / groups;
  const int g = co / Og;
  float acc = bias ? bias[co] : 0.f;
  const float* wr = w + (long long)co * Cg * KS;
  for (int cl = 0; cl < Cg; ++cl) {
    const float* xr = x + ((long long)b * Cin + g * Cg + cl) * Tin;
    for (int k = 0; k < KS; ++k) {
      const int ti = t * stride + k - pad;
      if (ti >= 0 && ti < Tin) acc = fmaf(wr[cl * KS + k], xr[ti], acc);
    }
  }
  y[((long long)b * Cout + co) * Tout + t] = acc;
}
__global__ void gconv_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                   int Cin, int Cout, int Tin, int Tout, int KS, int stride, int pad, int groups) {
  const int ti = blockIdx.x * blockDim.x + threadIdx.x;
  const int ci = blockIdx.y, b = blockIdx.z;
  if (ti >= Tin) return;
  const int Cg = Cin / groups, Og = Cout / groups;
  const int g = ci / Cg, cl = ci % Cg;
  const int kfirst = (ti + pad) % stride, tfirst = (ti + pad - kfirst) / stride;
  float acc = 0.f;
  for (int ol = 0; ol < Og; ++ol) {
    const int co = g * Og + ol;
    const float* wr = w + ((long long)co * Cg + cl) * KS;
    const float* dr = dy + ((long long)b * Cout + co) * Tout;
    // only taps k == (ti + pad) mod stride reach an output sample: t = (ti + pad - k) / stride
    for (int k = kfirst, t = tfirst; k < KS && t >= 0; k += stride, --t)
      if (t < Tout) acc = fmaf(wr[k], dr[t], acc);
  }
  dx[((long long)b * Cin + ci) * Tin + ti] = acc;
}
// one block per (co, cl), threads over k then reduce over (b,t) in-thread
__global__ void gconv_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw,
                                   int B, int Cin, int Cout, int Tin, int Tout, int KS, int stride, int pad, int groups) {
  __shared__ double sh[256];
  const int k = blockIdx.x, cl = blockIdx.y, co = blockIdx.z;
  const int Cg = Cin / groups, Og = Cout / groups;
  const int g = co / Og;
  double acc = 0.0;
  for (int b = 0; b < B; ++b) {
    const float* dr = dy + ((long long)b * Cout + co) * Tout;
    const float* xr = x + ((long long)b * Cin + g * Cg + cl) * Tin;
    for (int t = threadIdx.x; t < Tout; t += blockDim.x) {
      const int ti = t * stride + k - pad;
      if (ti >= 0 && ti < Tin) acc += (double)dr[t] * (double)xr[ti];
    }
  }
  const double s = block_sum_d(acc, sh);
  if (threadIdx.x == 0) dw[((long long)co * Cg + cl) * KS + k] = (float)s;
}

// ---- grouped strided Conv1d, LDS-tiled (v2) --------------------------------------------------------------------------
// The first versions above run one thread per output with one global load per FMA (and, for the weight gradient, one
// workgroup per weight element that re-reads its whole dy / x rows): 315 / 233 / 691 us per launch on DiscriminatorS'
// layers, 8.5 ms of a B=16 training iteration for 0.03 TFLOP (profiles/r02_w_train_B16_kernel_stats_final_build.txt).
// v2 stages the operands of ONE group in LDS and lets every thread produce all Og outputs (forward), all Cg inputs (dgrad)
// or KPT taps of one (o, cl) pair (wgrad) from them: Og (resp. Cg, ~1) FMAs per LDS read, weights read as LDS broadcasts.
// Index maths validated against torch by a numpy emulation before the HIP transcription (round 2, no GPU minute left for
// iteration): staging maps and loops below follow it line by line.
int g_gconv_version = 2;
__device__ __forceinline__ int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

constexpr int GC_TT = 256;      // outputs (fwd) / inputs (dgrad) per workgroup
template <int OG>
__global__ __launch_bounds__(GC_TT) void gconv_fwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y, int Cin,
                                                           int Cout, int Tin, int Tout, int KS, int stride, int pad, int Cg,
                                                           int Qn) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* ws = gsm;                               // [Cg*KS][OG]
  float* xs = gsm + Cg * KS * OG;                // [Cg][stride][Qn]: input position pp = q*stride + r at [r][q]
  const int tid = threadIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int t0 = blockIdx.x * GC_TT, p0 = t0 * stride - pad;
  const int NP = (GC_TT - 1) * stride + KS;
  for (int idx = tid; idx < Cg * NP; idx += GC_TT) {
    const int cl = idx / NP, pp = idx - cl * NP;
    const int ti = p0 + pp;
    const float v = (ti >= 0 && ti < Tin) ? x[((long long)b * Cin + g * Cg + cl) * Tin + ti] : 0.f;
    const int q = pp / stride, r = pp - q * stride;
    xs[(cl * stride + r) * Qn + q] = v;
  }
  const int nw = Cg * KS;
  for (int idx = tid; idx < nw * OG; idx += GC_TT) {
    const int j = idx / OG, o = idx - j * OG;
    ws[idx] = w[(long long)(g * OG + o) * nw + j];
  }
  __syncthreads();
  float acc[OG];
#pragma unroll
  for (int o = 0; o < OG; ++o) acc[o] = bias ? bias[g * OG + o] : 0.f;
  for (int cl = 0; cl < Cg; ++cl)
    for (int r = 0; r < stride; ++r) {
      const float* xr = xs + (cl * stride + r) * Qn + tid;
      for (int q = 0, k = r; k < KS; ++q, k += stride) {
        const float xv = xr[q];
        const float4* w4 = reinterpret_cast<const float4*>(ws + (cl * KS + k) * OG);
#pragma unroll
        for (int o4 = 0; o4 < OG / 4; ++o4) {
          const float4 wv = w4[o4];
          acc[4 * o4 + 0] = fmaf(wv.x, xv, acc[4 * o4 + 0]);
          acc[4 * o4 + 1] = fmaf(wv.y, xv, acc[4 * o4 + 1]);
          acc[4 * o4 + 2] = fmaf(wv.z, xv, acc[4 * o4 + 2]);
          acc[4 * o4 + 3] = fmaf(wv.w, xv, acc[4 * o4 + 3]);
        }
      }
    }
  const int t = t0 + tid;
  if (t < Tout) {
#pragma unroll
    for (int o = 0; o < OG; ++o) y[((long long)b * Cout + g * OG + o) * Tout + t] = acc[o];
  }
}

template <int OG, int CG>
__global__ __launch_bounds__(GC_TT) void gconv_dgrad2_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                             float* __restrict__ dx, int Cin, int Cout, int Tin, int Tout,
                                                             int KS, int stride, int pad, int Qd) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* ws = gsm;                               // [OG][CG][KS]
  float* ds = gsm + OG * CG * KS;                // [OG][Qd]: dy at t = tlo + q
  const int tid = threadIdx.x, g = blockIdx.y, b = blockIdx.z;
  const int ti0 = blockIdx.x * GC_TT;
  const int tlo = floor_div(ti0 + pad - (KS - 1), stride);
  for (int idx = tid; idx < OG * Qd; idx += GC_TT) {
    const int o = idx / Qd, q = idx - o * Qd;
    const int t = tlo + q;
    ds[idx] = (t >= 0 && t < Tout) ? dy[((long long)b * Cout + g * OG + o) * Tout + t] : 0.f;
  }
  for (int idx = tid; idx < OG * CG * KS; idx += GC_TT) ws[idx] = w[(long long)g * OG * CG * KS + idx];
  __syncthreads();
  const int ti = ti0 + tid;
  const int kf = (ti + pad) % stride, tf = (ti + pad - kf) / stride;     // tap k = kf + j*stride reaches output t = tf - j
  float acc[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) acc[c] = 0.f;
  for (int o = 0; o < OG; ++o) {
    const float* dr = ds + o * Qd + (tf - tlo);
    const float* wr = ws + o * CG * KS;
    for (int j = 0, k = kf; k < KS; ++j, k += stride) {
      const float dv = dr[-j];
#pragma unroll
      for (int c = 0; c < CG; ++c) acc[c] = fmaf(wr[c * KS + k], dv, acc[c]);
    }
  }
  if (ti < Tin) {
#pragma unroll
    for (int c = 0; c < CG; ++c) dx[((long long)b * Cin + g * CG + c) * Tin + ti] = acc[c];
  }
}

constexpr int GW_TT = 64;       // output time steps per staged tile
constexpr int GW_KPT = 11;      // taps per thread
constexpr int GW_PD = GW_TT + 1;
template <int OG, int CG>
__global__ __launch_bounds__(256) void gconv_wgrad2_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                           float* __restrict__ dw, int B, int Cin, int Cout, int Tin, int Tout,
                                                           int KS, int stride, int pad, int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) float gsm[];
  float* ds = gsm;                               // [OG][GW_PD]
  const int XW = (GW_TT - 1) * stride + KS;
  float* xs = gsm + OG * GW_PD;                  // [CG][XW]
  const int tid = threadIdx.x, g = blockIdx.y;
  const int NKQ = (KS + GW_KPT - 1) / GW_KPT;
  const bool worker = tid < OG * CG * NKQ;
  const int pair = tid / NKQ, kq = tid - pair * NKQ;
  const int o = worker ? pair / CG : 0, cl = worker ? pair - (pair / CG) * CG : 0;
  const int kbase = kq * GW_KPT;
  float acc[GW_KPT];
#pragma unroll
  for (int i = 0; i < GW_KPT; ++i) acc[i] = 0.f;
  const int n_t = (Tout + GW_TT - 1) / GW_TT;
  const int total = B * n_t;
  for (int tile = blockIdx.x; tile < total; tile += n_chunks) {
    const int b = tile / n_t, t0 = (tile - b * n_t) * GW_TT;
    const int p0 = t0 * stride - pad;
    __syncthreads();                             // previous tile consumed
    for (int idx = tid; idx < OG * GW_TT; idx += 256) {
      const int oo = idx / GW_TT, tl = idx - oo * GW_TT;
      const int t = t0 + tl;
      ds[oo * GW_PD + tl] = t < Tout ? dy[((long long)b * Cout + g * OG + oo) * Tout + t] : 0.f;
    }
    for (int idx = tid; idx < CG * XW; idx += 256) {
      const int cc = idx / XW, pp = idx - cc * XW;
      const int ti = p0 + pp;
      xs[idx] = (ti >= 0 && ti < Tin) ? x[((long long)b * Cin + g * CG + cc) * Tin + ti] : 0.f;
    }
    __syncthreads();
    if (worker) {
      const float* dr = ds + o * GW_PD;
      const float* xr = xs + cl * XW + kbase;
      for (int tl = 0; tl < GW_TT; ++tl) {
        const float dv = dr[tl];
        const float* xp = xr + tl * stride;
#pragma unroll
        for (int i = 0; i < GW_KPT; ++i)
          if (kbase + i < KS) acc[i] = fmaf(dv, xp[i], acc[i]);
      }
    }
  }
  if (worker) {
#pragma unroll
    for (int i = 0; i < GW_KPT; ++i)
      if (kbase + i < KS) atomicAdd(dw + ((long long)(g * OG + o) * CG + cl) * KS + kbase + i, acc[i]);
  }
}

// ---- scalar reductions for the losses (modules/losses.py:4-58, train.py:202,206) --------------------------------
// out += scale * sum_i f(a_i, b_i [, c_i, d_i])   accumulated in double per block, atomically added (double)
__global__ void reduce_scalar_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                     const float* __restrict__ c, const float* __restrict__ d, long long n,
                                     double* __restrict__ out, double scale) {
  __shared__ double sh[256];
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float x = a[i];
    switch (op) {
      case SVC_RED_SUM: acc += x; break;
      case SVC_RED_ABS_DIFF: acc += fabsf(x - b[i]); break;
      case SVC_RED_SQ_DIFF: { const float e = x - b[i]; acc += (double)e * e; } break;
      case SVC_RED_SQ_ONE_MINUS: { const float e = 1.f - x; acc += (double)e * e; } break;
      case SVC_RED_SQ: acc += (double)x * x; break;
      case SVC_RED_KL: {   // a=z_p b=logs_q c=m_p d=logs_p  (masked by caller through e = mask in `b2`) -> see host
        const float zp = x, lq = b[i], mp = c[i], lp = d[i];
        acc += (double)(lp - lq - 0.5f + 0.5f * (zp - mp) * (zp - mp) * expf(-2.f * lp));
      } break;
      default: break;
    }
  }
  const double s = block_sum_d(acc, sh);
  if (threadIdx.x == 0) atomicAdd(out, s * scale);
}
__global__ void scalar_to_float_kernel(const double* __restrict__ in, float* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

// ---- fused AdamW (train.py:79-88: lr, betas=(0.8,0.99), eps=1e-9, weight_decay default 0.01) -------------------
// Hyper-parameters live in DEVICE memory (hyper = [lr, beta1, beta2, eps, weight_decay, step, grad_scale]) so that a
// captured hipGraph of the whole training step stays valid while the step count advances and the scheduler changes lr.
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long long n, const float* __restrict__ hyper) {
  const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], step = hyper[5],
              gscale = hyper[6];
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2 = (float)(1.0 - pow((double)b2, (double)step));
  const float sq_bc2 = sqrtf(bc2), step_size = lr / bc1, decay = 1.f - lr * wd;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i] * gscale;
    float pi = p[i] * decay;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / sq_bc2 + eps;
    pi -= step_size * (mi / denom);
    p[i] = pi;
  }
}
__global__ void adamw_advance_kernel(float* hyper) { hyper[5] += 1.f; }

static inline unsigned grid1d(long long n, int bs = 256, unsigned cap = 65535u * 8) {
  long long g = (n + bs - 1) / bs;
  if (g < 1) g = 1;
  return (unsigned)(g > cap ? cap : g);
}

}  // namespace

extern "C" {

int svc_weight_norm_fwd_f32(const float* v, const float* g, float* w, float* norm, int rows, int cols, void* stream) {
  SVC_REQUIRE(v && g && w && rows > 0 && cols > 0, "weight_norm_fwd: bad args");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, w, norm, cols);
  return svc::check_launch("weight_norm_fwd");
}

int svc_weight_norm_bwd_f32(const float* v, const float* g, const float* norm, const float* dw, float* dv, float* dg,
                            int rows, int cols, void* stream) {
  SVC_REQUIRE(v && g && norm && dw && dv && dg && rows > 0 && cols > 0, "weight_norm_bwd: bad args");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, v, g, norm, dw, dv, dg, cols);
  return svc::check_launch("weight_norm_bwd");
}

int svc_pack_conv1d_weight_T(const float* w, float* dst, int Cout, int Cin, int KS, int CinP, void* stream) {
  SVC_REQUIRE(w && dst && Cout > 0 && Cin > 0 && KS > 0 && CinP >= Cin, "pack_conv1d_T: bad args");
  const long long n = (long long)Cout * KS * CinP;
  hipLaunchKernelGGL(pack_conv1d_T_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, w, dst, Cout, Cin, KS, CinP);
  return svc::check_launch("pack_conv1d_T");
}

static int wmap_from(const svc_conv_weight_args& a, WMap& p) {
  p.kind = a.kind; p.R = a.R; p.C2 = a.C2; p.K = a.K; p.Od = a.Od; p.Id = a.Id; p.Kd = a.Kd; p.OdP = a.OdP; p.IdP = a.IdP;
  p.s = a.s; p.shift = a.shift;
  SVC_REQUIRE(a.R > 0 && a.C2 > 0 && a.K > 0 && a.Od > 0 && a.Id > 0 && a.Kd > 0 && a.OdP >= a.Od && a.IdP >= a.Id,
              "conv_weight: bad shape");
  if (a.kind == 0) {
    SVC_REQUIRE(a.Od == a.R && a.Id == a.C2 && a.Kd == a.K, "conv_weight: dense map needs Od=R, Id=C2, Kd=K");
  } else if (a.kind == 1) {
    SVC_REQUIRE(a.s >= 1 && a.shift >= 0 && a.Od == a.R && a.Id == a.s * a.C2 && (a.K - 1 + a.shift) / a.s < a.Kd,
                "conv_weight: strided map needs Od=R, Id=s*C2, (K-1+shift)/s < Kd");
  } else if (a.kind == 2) {
    SVC_REQUIRE(a.s >= 1 && a.Od == a.s * a.C2 && a.Id == a.R && a.Kd == (a.K + a.s - 1) / a.s,
                "conv_weight: transposed map needs Od=u*C2, Id=R, Kd=ceil(K/u)");
  } else {
    SVC_REQUIRE(false, "conv_weight: kind must be 0 (dense), 1 (strided) or 2 (transposed)");
  }
  return SVC_OK;
}

int svc_conv_weight_prep_f32(const svc_conv_weight_args* args, void* stream) {
  SVC_REQUIRE(args && args->v && args->wp, "conv_weight_prep: null tensor");
  SVC_REQUIRE(!args->g || args->norm, "conv_weight_prep: weight-normed weights need the norm output");
  WMap p;
  const int rc = wmap_from(*args, p);
  if (rc != SVC_OK) return rc;
  if (args->g) hipLaunchKernelGGL(conv_weight_norm_kernel, dim3(p.R), dim3(256), 0, (hipStream_t)stream, args->v, args->norm, p.C2 * p.K);
  hipLaunchKernelGGL(conv_weight_scatter_kernel, dim3(wp_blocks(p.R, p.C2, p.K)), dim3(256), 0, (hipStream_t)stream, args->v,
                     args->g, args->wp, args->wt, args->norm, p);
  return svc::check_launch("conv_weight_prep");
}

int svc_conv_weight_prep_blocks(int R, int C2, int K) {
  return (R > 0 && C2 > 0 && K > 0) ? wp_blocks(R, C2, K) : 0;
}

int svc_conv_weight_prep_multi_f32(const svc_conv_weight_args* host_args, const svc_conv_weight_args* dev_args,
                                   const int* dev_row_start, const int* dev_block_start, int n_plans, void* stream) {
  SVC_REQUIRE(host_args && dev_args && dev_row_start && dev_block_start && n_plans > 0, "conv_weight_prep_multi: null table");
  long long rows = 0, blocks = 0;
  bool any_norm = false;
  for (int i = 0; i < n_plans; ++i) {
    const svc_conv_weight_args& a = host_args[i];
    SVC_REQUIRE(a.v && a.wp, "conv_weight_prep_multi: null tensor in plan %d", i);
    SVC_REQUIRE(!a.g || a.norm, "conv_weight_prep_multi: weight-normed plan %d needs the norm output", i);
    WMap p;
    const int rc = wmap_from(a, p);
    if (rc != SVC_OK) return rc;
    rows += a.R;
    blocks += wp_blocks(a.R, a.C2, a.K);
    any_norm = any_norm || a.g != nullptr;
  }
  SVC_REQUIRE(rows < (1ll << 31) && blocks < (1ll << 31), "conv_weight_prep_multi: too many rows");
  if (any_norm)
    hipLaunchKernelGGL(conv_weight_norm_multi_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dev_args,
                       dev_row_start, n_plans);
  hipLaunchKernelGGL(conv_weight_scatter_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dev_args,
                     dev_block_start, n_plans);
  return svc::check_launch("conv_weight_prep_multi");
}

int svc_conv_weight_grad_f32(const svc_conv_weight_args* args, const float* dwd, float* dv, float* dg, void* stream) {
  SVC_REQUIRE(args && args->v && dwd && dv, "conv_weight_grad: null tensor");
  SVC_REQUIRE(!args->g || (args->norm && dg), "conv_weight_grad: weight-normed weights need norm and dg");
  WMap p;
  const int rc = wmap_from(*args, p);
  if (rc != SVC_OK) return rc;
  hipLaunchKernelGGL(conv_weight_grad_kernel, dim3(p.R), dim3(256), 0, (hipStream_t)stream, args->v, args->g, args->norm,
                     dwd, dv, dg, p);
  return svc::check_launch("conv_weight_grad");
}

int svc_reduce_bct_f32(const float* x, float* out, long long x_bs, long long x_cs, int B, int C, int T, int mode,
                       float beta, void* stream) {
  SVC_REQUIRE(x && out && B > 0 && C > 0 && T > 0 && (mode == 0 || mode == 1), "reduce_bct: bad args");
  hipLaunchKernelGGL(reduce_bct_kernel, dim3(mode == 0 ? C : B * C), dim3(256), 0, (hipStream_t)stream, x, out, x_bs, x_cs,
                     B, C, T, mode, beta);
  return svc::check_launch("reduce_bct");
}

int svc_reduce_c_f32(const float* x, const float* w, float* out, int B, int C, int T, void* stream) {
  SVC_REQUIRE(x && out && B > 0 && C > 0 && T > 0, "reduce_c: bad args");
  hipLaunchKernelGGL(reduce_c_kernel, dim3(svc::cdiv(T, 256), B), dim3(256), 0, (hipStream_t)stream, x, w, out, C, T);
  return svc::check_launch("reduce_c");
}

/* y = leaky_relu'(x) * dy + r : the backward of `xt = c1(lrelu(x)); x = c2(xt) + x` (a ResBlock pair's input, vdecoder/hifigan/
 * models.py:60-67) at the point where x's two consumers meet — one launch instead of the leaky-ReLU backward plus the autograd
 * engine's accumulation add. */
int svc_lrelu_bwd_add_f32(const float* dy, const float* x, const float* r, float* y, long long n, float slope, void* stream) {
  SVC_REQUIRE(dy && x && r && y && n > 0, "lrelu_bwd_add: bad args");
  hipLaunchKernelGGL(lrelu_bwd_add_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, dy, x, r, y, n, slope);
  return svc::check_launch("lrelu_bwd_add");
}

int svc_ew_f32(int op, const float* a, const float* b, float* y, long long n, float alpha, float beta, void* stream) {
  SVC_REQUIRE(a && y && n > 0, "ew: bad args");
  hipLaunchKernelGGL(ew_kernel, dim3(grid1d(n)), dim3(256), 0, (hipStream_t)stream, op, a, b, y, n, alpha, beta);
  return svc::check_launch("ew");
}

int svc_ew_bct_f32(int op, const float* x, const float* side, float* y, long long x_bs, long long x_cs, long long s_bs,
                   long long s_cs, long long s_ts, long long y_bs, long long y_cs, int B, int C, int T, float alpha,
                   float beta, void* stream) {
  SVC_REQUIRE(x && side && y && B > 0 && C > 0 && T > 0, "ew_bct: bad args");
  hipLaunchKernelGGL(ew_bct_kernel, dim3(svc::cdiv(T, 256), C, B), dim3(256), 0, (hipStream_t)stream, op, x, side, y, x_bs,
                     x_cs, s_bs, s_cs, s_ts, y_bs, y_cs, C, T, alpha, beta);
  return svc::check_launch("ew_bct");
}

int svc_gate_fwd_f32(const float* in, float* acts, int B, int H, int T, void* stream) {
  SVC_REQUIRE(in && acts && B > 0 && H > 0 && T > 0, "gate_fwd: bad args");
  hipLaunchKernelGGL(gate_fwd_kernel, dim3(svc::cdiv(T, 256), H, B), dim3(256), 0, (hipStream_t)stream, in, acts, H, T);
  return svc::check_launch("gate_fwd");
}

int svc_gate_bwd_f32(const float* in, const float* dacts, float* din, int B, int H, int T, void* stream) {
  SVC_REQUIRE(in && dacts && din && B > 0 && H > 0 && T > 0, "gate_bwd: bad args");
  hipLaunchKernelGGL(gate_bwd_kernel, dim3(svc::cdiv(T, 256), H, B), dim3(256), 0, (hipStream_t)stream, in, dacts, din, H, T);
  return svc::check_launch("gate_bwd");
}

int svc_decimate_f32(const float* x, float* y, int B, int C, int T, int s, int w, int off, int Q, int lp, void* stream) {
  SVC_REQUIRE(x && y && B > 0 && C > 0 && T > 0 && s >= 1 && w >= 1 && Q > 0 && lp >= T, "decimate: bad args");
  SVC_REQUIRE(lp - T <= T - 1, "decimate: reflect padding longer than the signal");
  hipLaunchKernelGGL(decimate_kernel, dim3(svc::cdiv(Q * w, 256), s * C, B), dim3(256), 0, (hipStream_t)stream, x, y, C, T,
                     s, w, off, Q, lp);
  return svc::check_launch("decimate");
}

int svc_decimate_bwd_f32(const float* dy, float* dx, int B, int C, int T, int s, int w, int off, int Q, int lp,
                         void* stream) {
  SVC_REQUIRE(dy && dx && B > 0 && C > 0 && T > 0 && s >= 1 && w >= 1 && Q > 0 && lp >= T, "decimate_bwd: bad args");
  hipLaunchKernelGGL(decimate_bwd_kernel, dim3(svc::cdiv(T, 256), C, B), dim3(256), 0, (hipStream_t)stream, dy, dx, C, T,
                     s, w, off, Q, lp);
  return svc::check_launch("decimate_bwd");
}

int svc_debug_set_gconv_version(int version) {
  if (version != 1 && version != 2) return SVC_ERR_BAD_ARG;
  g_gconv_version = version;
  return SVC_OK;
}

int svc_gconv1d_fwd_f32(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Tin,
                        int Tout, int KS, int stride, int pad, int groups, void* stream) {
  SVC_REQUIRE(x && w && y && groups >= 1 && Cin % groups == 0 && Cout % groups == 0, "gconv_fwd: bad args");
  const int Cg = Cin / groups, Og = Cout / groups;
  int Qn = GC_TT + (KS - 1) / stride + 1;
  Qn |= 1;
  const size_t lds = sizeof(float) * ((size_t)Cg * KS * Og + (size_t)Cg * stride * Qn);
  if (g_gconv_version == 2 && (Og == 4 || Og == 16) && lds <= 60 * 1024 && stride >= 1) {
    const dim3 grid(svc::cdiv(Tout, GC_TT), groups, B);
    if (Og == 16)
      hipLaunchKernelGGL(gconv_fwd2_kernel<16>, grid, dim3(GC_TT), lds, (hipStream_t)stream, x, w, bias, y, Cin, Cout, Tin, Tout,
                         KS, stride, pad, Cg, Qn);
    else
      hipLaunchKernelGGL(gconv_fwd2_kernel<4>, grid, dim3(GC_TT), lds, (hipStream_t)stream, x, w, bias, y, Cin, Cout, Tin, Tout,
                         KS, stride, pad, Cg, Qn);
    return svc::check_launch("gconv_fwd2");
  }
  hipLaunchKernelGGL(gconv_fwd_kernel, dim3(svc::cdiv(Tout, 64), Cout, B), dim3(64), 0, (hipStream_t)stream, x, w, bias, y,
                     Cin, Cout, Tin, Tout, KS, stride, pad, groups);
  return svc::check_launch("gconv_fwd");
}

int svc_gconv1d_dgrad_f32(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int Tin, int Tout, int KS,
                          int stride, int pad, int groups, void* stream) {
  SVC_REQUIRE(dy && w && dx && groups >= 1 && Cin % groups == 0 && Cout % groups == 0, "gconv_dgrad: bad args");
  const int Cg = Cin / groups, Og = Cout / groups;
  const int Qd = (GC_TT + KS - 2) / stride + 2;
  const size_t lds = sizeof(float) * ((size_t)Og * Cg * KS + (size_t)Og * Qd);
  if (g_gconv_version == 2 && (Og == 4 || Og == 16) && Cg == 4 && lds <= 60 * 1024) {
    const dim3 grid(svc::cdiv(Tin, GC_TT), groups, B);
    if (Og == 16)
      hipLaunchKernelGGL((gconv_dgrad2_kernel<16, 4>), grid, dim3(GC_TT), lds, (hipStream_t)stream, dy, w, dx, Cin, Cout, Tin, Tout,
                         KS, stride, pad, Qd);
    else
      hipLaunchKernelGGL((gconv_dgrad2_kernel<4, 4>), grid, dim3(GC_TT), lds, (hipStream_t)stream, dy, w, dx, Cin, Cout, Tin, Tout,
                         KS, stride, pad, Qd);
    return svc::check_launch("gconv_dgrad2");
  }
  hipLaunchKernelGGL(gconv_dgrad_kernel, dim3(svc::cdiv(Tin, 64), Cin, B), dim3(64), 0, (hipStream_t)stream, dy, w, dx, Cin,
                     Cout, Tin, Tout, KS, stride, pad, groups);
  return svc::check_launch("gconv_dgrad");
}

int svc_gconv1d_wgrad_f32(const float* dy, const float* x, float* dw, int B, int Cin, int Cout, int Tin, int Tout, int KS,
                          int stride, int pad, int groups, void* stream) {
  SVC_REQUIRE(dy && x && dw && groups >= 1 && Cin % groups == 0 && Cout % groups == 0, "gconv_wgrad: bad args");
  const int Cg = Cin / groups, Og = Cout / groups;
  const int NKQ = (KS + GW_KPT - 1) / GW_KPT;
  const size_t lds = sizeof(float) * ((size_t)Og * GW_PD + (size_t)Cg * ((GW_TT - 1) * stride + KS));
  if (g_gconv_version == 2 && (Og == 4 || Og == 16) && Cg == 4 && Og * Cg * NKQ <= 256 && lds <= 60 * 1024) {
    if (hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * Cg * KS, (hipStream_t)stream) != hipSuccess) {
      svc::set_error("gconv_wgrad: memset failed");
      return SVC_ERR_HIP;
    }
    const int total = B * svc::cdiv(Tout, GW_TT);
    const int n_chunks = std::max(1, std::min(total, 1024 / groups));
    const dim3 grid(n_chunks, groups);
    if (Og == 16)
      hipLaunchKernelGGL((gconv_wgrad2_kernel<16, 4>), grid, dim3(256), lds, (hipStream_t)stream, dy, x, dw, B, Cin, Cout, Tin,
                         Tout, KS, stride, pad, n_chunks);
    else
      hipLaunchKernelGGL((gconv_wgrad2_kernel<4, 4>), grid, dim3(256), lds, (hipStream_t)stream, dy, x, dw, B, Cin, Cout, Tin,
                         Tout, KS, stride, pad, n_chunks);
    return svc::check_launch("gconv_wgrad2");
  }
  hipLaunchKernelGGL(gconv_wgrad_kernel, dim3(KS, Cin / groups, Cout), dim3(64), 0, (hipStream_t)stream, dy, x, dw, B, Cin,
                     Cout, Tin, Tout, KS, stride, pad, groups);
  return svc::check_launch("gconv_wgrad");
}

int svc_reduce_scalar_f64(int op, const float* a, const float* b, const float* c, const float* d, long long n,
                          double* out, double scale, void* stream) {
  SVC_REQUIRE(a && out && n > 0, "reduce_scalar: bad args");
  hipLaunchKernelGGL(reduce_scalar_kernel, dim3(grid1d(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, op, a, b, c, d, n,
                     out, scale);
  return svc::check_launch("reduce_scalar");
}

int svc_f64_to_f32(const double* in, float* out, int n, void* stream) {
  SVC_REQUIRE(in && out && n > 0, "f64_to_f32: bad args");
  hipLaunchKernelGGL(scalar_to_float_kernel, dim3(svc::cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, in, out, n);
  return svc::check_launch("f64_to_f32");
}

int svc_adamw_f32(float* p, const float* g, float* m, float* v, long long n, const float* hyper, void* stream) {
  SVC_REQUIRE(p && g && m && v && hyper && n > 0, "adamw: bad args");
  hipLaunchKernelGGL(adamw_kernel, dim3(grid1d(n, 256, 256 * 16)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, hyper);
  return svc::check_launch("adamw");
}

int svc_adamw_advance(float* hyper, void* stream) {
  SVC_REQUIRE(hyper != nullptr, "adamw_advance: null hyper");
  hipLaunchKernelGGL(adamw_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, hyper);
  return svc::check_launch("adamw_advance");
}

}  // extern "C"
