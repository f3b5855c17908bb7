/* oracle/hash_oracle.c — plain C restatement of the integer / ordering-critical pieces of the nerfacto hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle/README.md): the product never links or calls this.
 *
 *  - hash corner indices of the torch-path HashEncoding      (field_components/encodings.py:398-415, 417-438)
 *  - the PDF sampler's CDF, scanned left-to-right with a DOUBLE accumulator and fp32 outputs — what ATen's CPU
 *    cumsum does for fp32 tensors (acc_type<float> = double)      (model_components/ray_samplers.py:303-313)
 *  - torch.searchsorted(side="right") over that CDF           (model_components/ray_samplers.py:341)
 * Built with -O2 -ffp-contract=off so that no product-sum is fused: every fp32 operation rounds once, like the
 * eager torch ops it restates. tests/test_oracle_vs_golden.py cross-checks it against the numpy/torch oracle and
 * the reference fixtures.
 */
#include <math.h>
#include <stdint.h>

#define PRIME_Y 2654435761u
#define PRIME_Z 805459861u

/* x: [M,3] in [0,1]; out: [M,8] int64 table rows (corner bit0 = x ceil, bit1 = y ceil, bit2 = z ceil) */
void oracle_hash_corner_indices(const float* x, int64_t M, float scale, int level, int log2_table_size, int64_t* out) {
  const uint32_t mask = (1u << log2_table_size) - 1u;
  for (int64_t p = 0; p < M; ++p) {
    int32_t lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
      const float s = x[3 * p + a] * scale;
      lo[a] = (int32_t)floorf(s);
      hi[a] = (int32_t)ceilf(s);
    }
    for (int c = 0; c < 8; ++c) {
      const uint32_t ix = (uint32_t)((c & 1) ? hi[0] : lo[0]);
      const uint32_t iy = (uint32_t)((c & 2) ? hi[1] : lo[1]);
      const uint32_t iz = (uint32_t)((c & 4) ? hi[2] : lo[2]);
      const uint32_t h = (ix ^ (iy * PRIME_Y) ^ (iz * PRIME_Z)) & mask;
      out[8 * p + c] = (int64_t)h + ((int64_t)level << log2_table_size);
    }
  }
}

/* weights [N,S] -> cdf [N,S+1] */
void oracle_pdf_cdf(const float* weights, int64_t N, int S, float hist_pad, float eps, float* cdf) {
  for (int64_t r = 0; r < N; ++r) {
    const float* w = weights + r * S;
    float* c = cdf + r * (S + 1);
    double acc = 0.0;
    for (int i = 0; i < S; ++i) acc = acc + (double)(w[i] + hist_pad);
    float sum = (float)acc;
    float pad = eps - sum;
    if (!(pad > 0.0f)) pad = 0.0f;
    const float per = pad / (float)S;
    sum = sum + pad;
    double run = 0.0;
    c[0] = 0.0f;
    for (int i = 0; i < S; ++i) {
      const float pdf = ((w[i] + hist_pad) + per) / sum;
      run = run + (double)pdf;
      const float r = (float)run;
      c[i + 1] = r < 1.0f ? r : 1.0f;
    }
  }
}

/* number of entries of each sorted row a[r, 0..n) that are <= v[r, j]  (searchsorted side="right") */
void oracle_searchsorted_right(const float* a, int64_t N, int n, const float* v, int m, int32_t* out) {
  for (int64_t r = 0; r < N; ++r)
    for (int j = 0; j < m; ++j) {
      int lo = 0, hi = n;
      const float key = v[r * m + j];
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[r * n + mid] <= key) lo = mid + 1;
        else hi = mid;
      }
      out[r * m + j] = lo;
    }
}
