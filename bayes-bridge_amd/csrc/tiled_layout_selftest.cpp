// Self-test of the tiled layout builder + emulator for the sanitizer builds
// (make sanitize: -fsanitize=address,undefined and -fsanitize=thread; the
// builder is multi-threaded over panels).  Generates CSR matrices with an LCG,
// builds the layout with several worker threads (value-free and valued),
// emulates the kernel's walk and compares with a plain CSR product.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tiled_layout.hpp"

namespace {

struct Lcg {
  uint64_t s;
  uint32_t next() {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    return (uint32_t)(s >> 33);
  }
  double unit() { return (next() + 0.5) / 2147483648.0; }
};

int run_case(int64_t R, int64_t C, double density, bool binary,
             int force_PR, int force_G, uint64_t seed, int packed = -1) {
  Lcg g{seed};
  std::vector<int64_t> rowptr((size_t)R + 1, 0);
  std::vector<int32_t> colidx;
  std::vector<double> vals;
  for (int64_t r = 0; r < R; ++r) {
    // skewed columns (hot ones recur), ascending, duplicates allowed
    int64_t c = 0;
    const double row_density = density * (0.25 + 1.5 * g.unit());
    while (true) {
      const double u = g.unit();
      c += 1 + (int64_t)(-std::log(u) / row_density * (c < C / 8 ? 0.2 : 1.0));
      if (c >= C) break;
      colidx.push_back((int32_t)c);
      vals.push_back(binary ? 1.0 : g.unit() - 0.5);
      if (g.unit() < 0.01) {  // duplicate entry
        colidx.push_back((int32_t)c);
        vals.push_back(binary ? 1.0 : g.unit() - 0.5);
      }
    }
    rowptr[(size_t)r + 1] = (int64_t)colidx.size();
  }
  const int64_t nnz = (int64_t)colidx.size();
  std::vector<double> x((size_t)C), ref((size_t)R, 0.);
  for (auto& v : x) v = g.unit() - 0.5;
  for (int64_t r = 0; r < R; ++r)
    for (int64_t k = rowptr[(size_t)r]; k < rowptr[(size_t)r + 1]; ++k)
      ref[(size_t)r] += vals[(size_t)k] * x[(size_t)colidx[(size_t)k]];
  bbx::TiledOptions opt;
  opt.force_PR = force_PR;
  opt.force_G = force_G;
  opt.max_threads = 4;
  opt.packed = packed;
  bbx::TiledHost m;
  std::string err;
  if (colidx.empty()) colidx.push_back(0);
  if (bbx::build_tiled_host(R, C, nnz, rowptr.data(), colidx.data(),
                            binary ? nullptr : vals.data(), opt, &m, &err) != 0) {
    fprintf(stderr, "build failed: %s\n", err.c_str());
    return 1;
  }
  std::vector<double> slab;
  bbx::emulate_tiled_spmv(m, x.data(), &slab);
  double worst = 0.;
  for (int64_t r = 0; r < R; ++r) {
    double a = 0.;
    for (int gq = 0; gq < m.G; ++gq) a += slab[(size_t)gq * (size_t)R + (size_t)r];
    worst = std::fmax(worst, std::fabs(a - ref[(size_t)r]));
  }
  const double cyc = bbx::tiled_mean_gather_cycles(m);
  printf("R=%lld C=%lld nnz=%lld %s%s PR=%d G=%d W=%d blocks=%d extras=%d: "
         "max err %.2e, %.2f LDS cycles per gather\n",
         (long long)R, (long long)C, (long long)nnz, binary ? "binary" : "valued",
         m.packed ? " packed" : "", m.PR, m.G, m.W, m.n_block, m.n_extra, worst, cyc);
  return worst <= 1e-10 ? 0 : 1;
}

// The 64-bit constructor's host utilities: the threaded transposition against
// a serial one, the transposed matrix through the layout + emulator, and the
// structure check on a broken copy.
int run_transpose_case(int64_t R, int64_t C, double density, uint64_t seed) {
  Lcg g{seed};
  std::vector<int64_t> rowptr((size_t)R + 1, 0), col64;
  std::vector<int32_t> colidx;
  std::vector<double> vals;
  for (int64_t r = 0; r < R; ++r) {
    for (int64_t c = 0; c < C; ++c)
      if (g.unit() < density) {
        colidx.push_back((int32_t)c);
        col64.push_back(c);
        vals.push_back(g.unit() - .5);
      }
    rowptr[(size_t)r + 1] = (int64_t)colidx.size();
  }
  const int64_t nnz = (int64_t)colidx.size();
  if (nnz == 0) return 0;
  if (bbx::check_csr64_host(R, C, nnz, rowptr.data(), col64.data(), 8) != 0) return 1;
  bbx::HostCsr t;
  bbx::transpose_csr_host(R, C, rowptr.data(), colidx.data(), vals.data(), 8, &t);
  // serial reference: entries of column j in row order
  std::vector<int64_t> at((size_t)C + 1, 0);
  for (int64_t k = 0; k < nnz; ++k) at[(size_t)colidx[(size_t)k] + 1] += 1;
  for (int64_t j = 0; j < C; ++j) at[(size_t)j + 1] += at[(size_t)j];
  int bad = 0;
  for (int64_t j = 0; j <= C; ++j) bad += t.rowptr[(size_t)j] != at[(size_t)j];
  std::vector<int64_t> pos(at.begin(), at.end() - 1);
  for (int64_t r = 0; r < R; ++r)
    for (int64_t k = rowptr[(size_t)r]; k < rowptr[(size_t)r + 1]; ++k) {
      const int64_t w = pos[(size_t)colidx[(size_t)k]]++;
      bad += t.colidx[(size_t)w] != (int32_t)r || t.vals[(size_t)w] != vals[(size_t)k];
    }
  // broken copies: a column id out of range, a row out of order
  std::vector<int64_t> broken = col64;
  broken[(size_t)(nnz / 2)] = C;
  bad += (bbx::check_csr64_host(R, C, nnz, rowptr.data(), broken.data(), 8) & 2) == 0;
  if (rowptr[1] >= 2) {
    broken = col64;
    std::swap(broken[0], broken[1]);
    bad += (bbx::check_csr64_host(R, C, nnz, rowptr.data(), broken.data(), 8) & 4) == 0;
  }
  printf("transpose R=%lld C=%lld nnz=%lld: %s\n", (long long)R, (long long)C,
         (long long)nnz, bad ? "MISMATCH" : "equal to the serial transposition");
  return bad ? 1 : 0;
}

}  // namespace

int main() {
  int bad = 0;
  bad += run_case(3000, 900, .05, true, 0, 0, 1);
  bad += run_case(3000, 900, .05, false, 256, 0, 2);
  bad += run_case(700, 40000, .002, true, 128, 2, 3);
  bad += run_case(700, 40000, .002, true, 128, 3, 4);
  bad += run_case(5000, 17000, .004, false, 512, 2, 5);
  bad += run_case(17, 3, .6, true, 0, 0, 6);
  bad += run_case(1, 70000, .001, true, 0, 0, 7);
  // packed groups forced on and off (the builder picks by step count)
  bad += run_case(3000, 900, .05, true, 0, 0, 1, 1);
  bad += run_case(3000, 900, .05, true, 0, 0, 1, 0);
  bad += run_case(700, 40000, .002, true, 128, 2, 3, 1);
  bad += run_case(900, 33000, .0004, true, 0, 0, 8, 1);  // gaps beyond 4095 slots
  bad += run_case(1, 70000, .001, true, 0, 0, 7, 1);
  bad += run_transpose_case(3000, 900, .05, 11);
  bad += run_transpose_case(5, 40000, .01, 12);
  bad += run_transpose_case(20000, 7, .3, 13);
  if (bad) fprintf(stderr, "%d case(s) FAILED\n", bad);
  return bad ? 1 : 0;
}
