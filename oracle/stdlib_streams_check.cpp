// Test tool (not product, not reference code): draws d normals then one uniform per
// iteration from the SAME three objects the reference constructs at
// adaptMCMC_source.cpp:78-80, using this toolchain's libstdc++, so that
// tests/test_oracle.py can pin orc_stdlib_streams() bit for bit.
#include <random>

extern "C" void stdcheck_streams(double seedMH, int n_iterations, int d, double *Z, double *U)
{
    std::mt19937 engine(seedMH);
    std::uniform_real_distribution<> uniform(0.0, 1.0);
    std::normal_distribution<> normal(0.0, 1.0);
    for (int i = 0; i < n_iterations; ++i) {
        for (int k = 0; k < d; ++k) Z[(long)i * d + k] = normal(engine);
        U[i] = uniform(engine);
    }
}
