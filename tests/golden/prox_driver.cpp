// Fixture generator (ours): evaluates the reference's ProxEval / FuncEval
// (src/include/prox_lib.h, included from where it lies via -I) on a grid and
// writes rows of 11 doubles [is_float, h, a, b, c, d, e, rho, v, prox, func] to stdout.
// Built and run by tests/golden/make_golden.py in the build container only.
#include <cstdio>
#include <vector>

#include "prox_lib.h"

template <typename T>
void run(double is_float, std::vector<double> *out) {
  const double coefs[3][5] = {{1.0, 0.0, 1.0, 0.0, 0.0}, {-1.5, 0.3, 2.0, -0.2, 0.5}, {0.7, -0.4, 0.25, 0.1, 0.0}};
  const double rhos[3] = {0.1, 1.0, 7.5};
  const double vs[16] = {-6.0, -3.2, -1.7, -1.0, -0.55, -0.2, -0.01, 0.0, 0.01, 0.3, 0.5, 1.0, 1.3, 2.6, 4.0, 9.5};
  for (int h = 0; h < 16; ++h)
    for (const auto &c : coefs)
      for (double rho : rhos)
        for (double v : vs) {
          FunctionObj<T> f(static_cast<Function>(h), static_cast<T>(c[0]), static_cast<T>(c[1]),
                           static_cast<T>(c[2]), static_cast<T>(c[3]), static_cast<T>(c[4]));
          const T p = ProxEval(f, static_cast<T>(v), static_cast<T>(rho));
          const T fe = FuncEval(f, static_cast<T>(v));
          const double row[11] = {is_float, (double)h, c[0], c[1], c[2], c[3], c[4], rho, v, (double)p, (double)fe};
          out->insert(out->end(), row, row + 11);
        }
}

int main() {
  std::vector<double> out;
  run<double>(0.0, &out);
  run<float>(1.0, &out);
  fwrite(out.data(), sizeof(double), out.size(), stdout);
  return 0;
}
