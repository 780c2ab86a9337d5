// GEMM lab: standalone timing of the dense-layer kernels at the train step's shapes (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DNSKY_LAB_*] tools/gemm_lab/lab.cpp -o build/lab_x && build/lab_x
#include "../../neusky_amd/csrc/api.cpp"
#include "../../neusky_amd/csrc/gemm_f32.hip"
#include <stdio.h>
#include <vector>
#include <random>
#include <math.h>

static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  std::mt19937 g(seed);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  for (auto& x : h) x = u(g) * scale;
  float* d;
  hipMalloc(&d, n * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  return d;
}

static double time_gemm(nsky_gemm_desc d, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) if (nsky_gemm_f32(&d, 0)) { printf("ERR %s\n", nsky_last_error()); return -1; }
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) nsky_gemm_f32(&d, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / iters;
}

int main(int argc, char** argv) {
  const int M = 262144;
  float* X = dev_rand((size_t)M * 2560, 1.0f, 1);     // activations / gradients
  float* W = dev_rand((size_t)2560 * 256, 1.f / 16, 2);
  float* Cb = dev_rand((size_t)M * 2560, 1.0f, 3);
  float* aux = dev_rand((size_t)M * 256 * 3, 1.0f, 4);
  struct Case { const char* name; int M, N, K, akc, bkc, prec, ks, epi; } cases[] = {
    {"F  nt H    262144x256x256 ", M, 256, 256, 1, 1, NSKY_PREC_F16X2, 1, NSKY_EPI_NONE},
    {"F  nt H film                ", M, 256, 256, 1, 1, NSKY_PREC_F16X2, 1, NSKY_EPI_FILM},
    {"F  nt H    262144x2560x256", M, 2560, 256, 1, 1, NSKY_PREC_F16X2, 1, NSKY_EPI_NONE},
    {"D  nn b2   262144x256x256 ", M, 256, 256, 1, 0, NSKY_PREC_BF16X2, 1, NSKY_EPI_NONE},
    {"D  nn b2   262144x256x2560", M, 256, 2560, 1, 0, NSKY_PREC_BF16X2, 1, NSKY_EPI_NONE},
    {"W  tn b2   256x256x262144 ", 256, 256, M, 0, 0, NSKY_PREC_BF16X2, 128, NSKY_EPI_NONE},
    {"W  tn b2   2560x256x262144", 2560, 256, M, 0, 0, NSKY_PREC_BF16X2, 76, NSKY_EPI_NONE},
  };
  if (argc > 1 && argv[1][0] == 'p') goto planes;
  for (auto& c : cases) {
    nsky_gemm_desc d = {};
    d.A = X; d.B = (c.akc && !c.bkc) || c.akc ? W : X + (size_t)M * 256; d.C = Cb;
    d.M = c.M; d.N = c.N; d.K = c.K;
    d.a_kcontig = c.akc; d.b_kcontig = c.bkc;
    d.lda = c.akc ? c.K : c.M; d.ldb = c.bkc ? c.K : c.N; d.ldc = c.N;
    if (!c.akc) { d.A = X; d.lda = c.M; d.B = X + (size_t)M * 2560 / 2; d.ldb = c.N; }
    d.epi = c.epi; d.precision = c.prec; d.k_splits = c.ks;
    if (c.epi == NSKY_EPI_FILM) { d.aux0 = aux; d.ldaux0 = 256; d.aux1 = aux + (size_t)M * 256; d.ldaux1 = 256; d.out1 = aux + (size_t)M * 512; d.ldout1 = 256; d.p0 = 15.f; d.p1 = 30.f; }
    double ms = time_gemm(d, 10);
    printf("%s %8.1f us  %7.1f TF/s\n", c.name, ms * 1e3, 2.0 * c.M * c.N * c.K / ms / 1e9);
  }
planes:
  // ---- LDS-DMA planes kernel vs the register-staged kernel on the same operands
  uint16_t* planes = nullptr; hipMalloc(&planes, (size_t)2 * 2560 * 2560 * 2);
  float* Cref = nullptr; hipMalloc(&Cref, (size_t)M * 2560 * 4);
  struct PCase { const char* name; int M, N, K, transpose, prec, epi; } pc[] = {
    {"F  planes H  262144x256x256 ", M, 256, 256, 0, NSKY_PREC_F16X2, NSKY_EPI_NONE},
    {"F  planes H film            ", M, 256, 256, 0, NSKY_PREC_F16X2, NSKY_EPI_FILM},
    {"F  planes H  262144x2560x256", M, 2560, 256, 0, NSKY_PREC_F16X2, NSKY_EPI_NONE},
    {"D  planes b2 262144x256x256 ", M, 256, 256, 1, NSKY_PREC_BF16X2, NSKY_EPI_NONE},
    {"D  planes b2 262144x256x2560", M, 256, 2560, 1, NSKY_PREC_BF16X2, NSKY_EPI_NONE},
    {"F  planes H  153600x128x128 ", 153600, 128, 128, 0, NSKY_PREC_F16X2, NSKY_EPI_NONE},
    {"F  planes H  1000x300x256   ", 1000, 300, 256, 0, NSKY_PREC_F16X2, NSKY_EPI_RELU},
    {"D  planes b2 153600x128x128 bwdfilm", 153600, 128, 128, 1, NSKY_PREC_BF16X2, NSKY_EPI_BWD_FILM},
    {"D  planes b2 153600x128x128 plain  ", 153600, 128, 128, 1, NSKY_PREC_BF16X2, NSKY_EPI_NONE},
    {"D  planes b2 264448x256x256 bwdfilm", M, 256, 256, 1, NSKY_PREC_BF16X2, NSKY_EPI_BWD_FILM},
  };
  for (auto& c : pc) {
    nsky_gemm_desc d = {};
    d.A = X; d.B = W; d.C = Cref; d.M = c.M; d.N = c.N; d.K = c.K;
    d.a_kcontig = 1; d.b_kcontig = !c.transpose; d.lda = c.K; d.ldb = c.transpose ? c.N : c.K; d.ldc = c.N;
    d.epi = c.epi; d.precision = c.prec;
    if (c.epi == NSKY_EPI_FILM) { d.aux0 = aux; d.ldaux0 = 256; d.aux1 = aux + (size_t)M * 256; d.ldaux1 = 256; d.out1 = aux + (size_t)M * 512; d.ldout1 = 256; d.p0 = 15.f; d.p1 = 30.f; }
    if (c.epi == NSKY_EPI_BWD_FILM) { d.aux0 = aux; d.ldaux0 = 256; d.aux1 = aux + (size_t)M * 256; d.ldaux1 = 256; d.aux2 = aux + (size_t)M * 512; d.ldaux2 = 256;
                                      d.out1 = X + (size_t)M * 1024; d.ldout1 = 256; d.out2 = X + (size_t)M * 1536; d.ldout2 = 256; d.p0 = 15.f; d.p1 = 30.f; }
    {
      hipEvent_t r0, r1; hipEventCreate(&r0); hipEventCreate(&r1);
      for (int i = 0; i < 2; ++i) nsky_gemm_f32(&d, 0);
      hipEventRecord(r0, 0);
      for (int i = 0; i < 10; ++i) nsky_gemm_f32(&d, 0);
      hipEventRecord(r1, 0); hipEventSynchronize(r1);
      float rms; hipEventElapsedTime(&rms, r0, r1);
      printf("   [register-staged kernel: %8.1f us] ", rms * 100);
    }
    if (nsky_gemm_f32(&d, 0)) { printf("ERR %s\n", nsky_last_error()); return 1; }
    const int rows_pad = (c.N + 255) / 256 * 256, ldp = c.K;
    uint16_t* hi = planes; uint16_t* lo = planes + (size_t)rows_pad * ldp;
    if (nsky_split_planes(W, c.N, c.K, c.transpose ? c.N : c.K, c.transpose, c.prec, hi, lo, rows_pad, ldp, 0)) { printf("ERR %s\n", nsky_last_error()); return 1; }
    d.C = Cb;
    hipMemset(Cb, 0xff, (size_t)c.M * c.N * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (nsky_gemm_f32_planes(&d, hi, lo, ldp, 0)) { printf("ERR %s\n", nsky_last_error()); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    for (int i = 0; i < 10; ++i) nsky_gemm_f32_planes(&d, hi, lo, ldp, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    // compare a sample of rows
    const size_t n = (size_t)c.M * c.N;
    std::vector<float> a(n > (1u << 24) ? (1u << 24) : n), b(a.size());
    hipMemcpy(a.data(), Cref, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), Cb, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0, mx = 0; size_t bad = 0;
    for (size_t i = 0; i < a.size(); ++i) { double dd = fabs((double)a[i] - b[i]); if (!(dd <= 1e30)) ++bad; if (dd > md) md = dd; if (fabs(a[i]) > mx) mx = fabs(a[i]); }
    // tail of the matrix too
    std::vector<float> ta(c.N * 128), tb(c.N * 128);
    hipMemcpy(ta.data(), Cref + (size_t)(c.M - 128) * c.N, ta.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(tb.data(), Cb + (size_t)(c.M - 128) * c.N, tb.size() * 4, hipMemcpyDeviceToHost);
    double mdt = 0; for (size_t i = 0; i < ta.size(); ++i) { double dd = fabs((double)ta[i] - tb[i]); if (!(dd <= 1e30)) ++bad; if (dd > mdt) mdt = dd; }
    printf("%s %8.1f us  %7.1f TF/s   maxdiff %.3g (tail %.3g) of max %.3g, nan/inf %zu\n", c.name, ms * 1e3, 2.0 * c.M * c.N * c.K / ms / 1e9, md, mdt, mx, bad);
  }
  return 0;
}
