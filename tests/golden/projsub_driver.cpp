// Fixture generator (ours): evaluates the reference's ProjSubgradEval
// (src/include/prox_lib.h:468-493, included from where it lies via -I) on a grid and writes rows of
// 10 doubles [is_float, h, a, b, c, d, e, x, v, result] to stdout.  Built and run by
// tests/golden/make_golden.py (projsub_table) in the build container only.
#include <cstdio>
#include <vector>

#include "prox_lib.h"

template <typename T>
void run(double is_float, std::vector<double> *out) {
  // the third set has a = 0 (the affine shortcut, :471-472), the fourth c = 0
  const double coefs[4][5] = {{1.0, 0.0, 1.0, 0.0, 0.0}, {-1.5, 0.3, 2.0, -0.2, 0.5}, {0.0, -0.4, 0.25, 0.1, 0.3},
                              {0.7, 0.2, 0.0, -0.3, 0.6}};
  // points: the kinks (a x - b = 0 or 1 for the first two sets) and both sides of them
  const double xs[11] = {-2.0, -0.2, 0.0, 0.2, 1.0, 0.5, -0.1, 1.7, -0.8666666666666667, 0.3, 2.5};
  const double vs[7] = {-3.0, -1.0, -0.4, 0.0, 0.6, 1.0, 2.5};
  for (int h = 0; h < 16; ++h)
    for (const auto &c : coefs)
      for (double x : xs)
        for (double v : vs) {
          FunctionObj<T> f(static_cast<Function>(h), static_cast<T>(c[0]), static_cast<T>(c[1]),
                           static_cast<T>(c[2]), static_cast<T>(c[3]), static_cast<T>(c[4]));
          const T r = ProjSubgradEval(f, static_cast<T>(v), static_cast<T>(x));
          const double row[10] = {is_float, (double)h, c[0], c[1], c[2], c[3], c[4], x, v, (double)r};
          out->insert(out->end(), row, row + 10);
        }
}

int main() {
  std::vector<double> out;
  run<double>(0.0, &out);
  run<float>(1.0, &out);
  fwrite(out.data(), sizeof(double), out.size(), stdout);
  return 0;
}
