#!/bin/bash
# Do the reference's OWN tests still compile against include/spmv/*.h?  
# Build container only: reads /root/reference/tests/*.c* where they lie, copies nothing, writes nothing but a
# throw-away gtest stand-in under $TMPDIR (assertion macros that type-check their arguments and stream operands; no
# test runs — the restated cases that RUN are tests/cpp/reference_suite.cpp).  -fsyntax-only, one file at a time.
# Exit code 0 = every file compiles.  Skips (exit 0, says so) where /root/reference is absent (the GPU box).
set -u
REF=${REFERENCE_ROOT:-/root/reference}
REPO="$(cd "$(dirname "$0")/.." && pwd)"
if [ ! -d "$REF/tests" ]; then echo "reference tests not present at $REF/tests: skipped"; exit 0; fi
SHIM="$(mktemp -d "${TMPDIR:-/tmp}/gtest_shim.XXXXXX")"
trap 'rm -rf "$SHIM"' EXIT
mkdir -p "$SHIM/gtest"
cat > "$SHIM/gtest/gtest.h" <<'SHIM_EOF'
// stand-in for <gtest/gtest.h>: enough surface for -fsyntax-only (fixtures, TEST / TEST_F, the assertion macros as
// type-checked expressions that accept << messages).  Not a test runner.
#pragma once
#include <cmath>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>
namespace testing {
class Test { public: virtual ~Test() {} protected: virtual void SetUp() {} virtual void TearDown() {} virtual void TestBody() = 0; };
struct Message { template <typename T> Message& operator<<(const T&) { return *this; } };
struct Voidify { void operator&(const Message&) const {} };
inline void InitGoogleTest(int*, char**) {}
}
inline int RUN_ALL_TESTS() { return 0; }
#define GTEST_SHIM_CHECK_(cond) switch (0) case 0: default: if (cond) ; else ::testing::Voidify() & ::testing::Message()
#define GTEST_SHIM_FATAL_(cond) switch (0) case 0: default: if (cond) ; else return ::testing::Voidify() & ::testing::Message()
#define EXPECT_TRUE(c) GTEST_SHIM_CHECK_(static_cast<bool>(c))
#define EXPECT_FALSE(c) GTEST_SHIM_CHECK_(!static_cast<bool>(c))
#define EXPECT_EQ(a, b) GTEST_SHIM_CHECK_((a) == (b))
#define EXPECT_NE(a, b) GTEST_SHIM_CHECK_((a) != (b))
#define EXPECT_LT(a, b) GTEST_SHIM_CHECK_((a) < (b))
#define EXPECT_LE(a, b) GTEST_SHIM_CHECK_((a) <= (b))
#define EXPECT_GT(a, b) GTEST_SHIM_CHECK_((a) > (b))
#define EXPECT_GE(a, b) GTEST_SHIM_CHECK_((a) >= (b))
#define EXPECT_FLOAT_EQ(a, b) GTEST_SHIM_CHECK_(static_cast<float>(a) == static_cast<float>(b))
#define EXPECT_NEAR(a, b, tol) GTEST_SHIM_CHECK_(std::fabs(static_cast<double>(a) - static_cast<double>(b)) <= static_cast<double>(tol))
#define EXPECT_STREQ(a, b) GTEST_SHIM_CHECK_(std::strcmp((a), (b)) == 0)
#define EXPECT_NO_THROW(stmt) switch (0) case 0: default: if (([&] { try { stmt; } catch (...) { return false; } return true; })()) ; else ::testing::Voidify() & ::testing::Message()
#define ASSERT_TRUE(c) GTEST_SHIM_FATAL_(static_cast<bool>(c))
#define ASSERT_EQ(a, b) GTEST_SHIM_FATAL_((a) == (b))
#define ASSERT_NE(a, b) GTEST_SHIM_FATAL_((a) != (b))
#define GTEST_SHIM_NAME_(suite, name) suite##_##name##_Test
#define TEST(suite, name) class GTEST_SHIM_NAME_(suite, name) : public ::testing::Test { void TestBody() override; }; void GTEST_SHIM_NAME_(suite, name)::TestBody()
#define TEST_F(fixture, name) class GTEST_SHIM_NAME_(fixture, name) : public fixture { void TestBody() override; }; void GTEST_SHIM_NAME_(fixture, name)::TestBody()
SHIM_EOF
status=0
for f in "$REF"/tests/*.cpp "$REF"/tests/*.cu; do
    [ -e "$f" ] || continue
    if g++ -std=c++17 -fsyntax-only -x c++ -D__HIP_PLATFORM_AMD__=1 -I"$SHIM" -I"$REPO/include" -I/opt/rocm/include "$f" 2> "$SHIM/err.txt"; then
        echo "ok      $(basename "$f")"
    else
        echo "FAILED  $(basename "$f")"; head -20 "$SHIM/err.txt"; status=1
    fi
done
exit $status
