// spmv/test_utils.h — seeded generators and tolerant comparators for tests.
//
// Names and signatures as in the reference (include/spmv/test_utils.h:12-79)
// so its test sources compile against this header unchanged.
#ifndef SPMV_TEST_UTILS_H
#define SPMV_TEST_UTILS_H

#include <algorithm>
#include <cmath>
#include <random>
#include <vector>

namespace spmv {
namespace test {

class RandomGenerator {
public:
    RandomGenerator(unsigned seed = 42) : engine_(seed) {}

    int randInt(int min, int max) {
        return std::uniform_int_distribution<int>(min, max)(engine_);
    }

    float randFloat(float min, float max) {
        return std::uniform_real_distribution<float>(min, max)(engine_);
    }

    bool randBool(float probability = 0.5f) {
        return randFloat(0.0f, 1.0f) < probability;
    }

private:
    std::mt19937 engine_;
};

// Row-major rows x cols matrix; each entry is non-zero with probability `density`.
inline std::vector<float> generateRandomDenseMatrix(
        int rows, int cols, float density, RandomGenerator& rng,
        float min_val = -10.0f, float max_val = 10.0f) {
    const size_t total = static_cast<size_t>(rows) * cols;
    std::vector<float> matrix(total, 0.0f);
    for (size_t i = 0; i < total; ++i) {
        if (rng.randBool(density)) matrix[i] = rng.randFloat(min_val, max_val);
    }
    return matrix;
}

inline std::vector<float> generateRandomVector(
        int size, RandomGenerator& rng,
        float min_val = -10.0f, float max_val = 10.0f) {
    std::vector<float> vec(size);
    for (float& v : vec) v = rng.randFloat(min_val, max_val);
    return vec;
}

inline bool floatArraysEqual(const float* a, const float* b, int size,
                             float abs_tol = 1e-6f, float rel_tol = 1e-6f) {
    for (int i = 0; i < size; ++i) {
        const float diff  = std::abs(a[i] - b[i]);
        const float scale = std::max(std::abs(a[i]), std::abs(b[i]));
        if (diff > abs_tol && diff > rel_tol * scale) return false;
    }
    return true;
}

inline bool intArraysEqual(const int* a, const int* b, int size) {
    return std::equal(a, a + size, b);
}

} // namespace test
} // namespace spmv

#endif // SPMV_TEST_UTILS_H
