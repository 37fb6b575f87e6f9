// TEST INFRASTRUCTURE -- a minimal stand-in for <gtest/gtest.h> (GoogleTest is not in this image) with exactly the macros the
// reference's own test sources use (TEST, ASSERT_* / EXPECT_*, InitGoogleTest, RUN_ALL_TESTS), so that
// /root/reference/test/**/*.test.cpp compile UNCHANGED, from where they lie, against the ungar_amd facade headers
// (oracle/ref_tests/build_ref_tests.sh).  What is under test is the facade; nothing here is reference code.
#pragma once

#include <cmath>
#include <cstdio>
#include <functional>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

namespace testing {

struct Registry {
    struct Case {
        std::string name;
        std::function<void()> body;
    };
    std::vector<Case> cases;
    int failures = 0;       // failed assertions of the running case
    bool fatal = false;     // an ASSERT_* failed: the rest of the case is skipped
    static Registry& Get() {
        static Registry r;
        return r;
    }
};

struct Registrar {
    Registrar(const char* suite, const char* name, std::function<void()> body) { Registry::Get().cases.push_back({std::string(suite) + "." + name, std::move(body)}); }
};

/// Collects the `<< message` operands of a failed assertion and reports on destruction.
class Message {
  public:
    Message(bool failed, const char* file, int line, const std::string& what) : failed_(failed) {
        if (failed_) text_ << file << ":" << line << ": Failure\n  " << what;
    }
    Message(Message&& o) : failed_(o.failed_), text_(std::move(o.text_)) { o.failed_ = false; }
    ~Message() {
        if (failed_) {
            std::cout << text_.str() << std::endl;
            ++Registry::Get().failures;
        }
    }
    template <class T>
    Message& operator<<(const T& v) {
        if (failed_) text_ << v;
        return *this;
    }
    explicit operator bool() const { return failed_; }

  private:
    bool failed_;
    std::ostringstream text_;
};

/// `return Voidify() & message;` -- lets ASSERT_* leave a void test body after the message operands were streamed.
struct Voidify {
    void operator&(const Message&) const {}
};

inline void InitGoogleTest() {}
inline void InitGoogleTest(int*, char**) {}

inline int RunAllTests() {
    auto& r = Registry::Get();
    int failed = 0;
    for (auto& c : r.cases) {
        std::cout << "[ RUN      ] " << c.name << std::endl;
        r.failures = 0;
        c.body();
        std::cout << (r.failures ? "[  FAILED  ] " : "[       OK ] ") << c.name << std::endl;
        failed += r.failures ? 1 : 0;
    }
    std::cout << "[==========] " << r.cases.size() << " tests ran, " << failed << " failed." << std::endl;
    if (!failed) std::cout << "[  PASSED  ] " << r.cases.size() << " tests." << std::endl;
    return failed ? 1 : 0;
}

template <class A, class B>
std::string Describe(const char* ea, const char* eb, const char* op, const A&, const B&) {
    return std::string("Expected: (") + ea + ") " + op + " (" + eb + ")";
}

}  // namespace testing

#define RUN_ALL_TESTS() ::testing::RunAllTests()

#define TEST(suite, name)                                                                                   \
    static void suite##_##name##_Body();                                                                    \
    static ::testing::Registrar suite##_##name##_registrar(#suite, #name, &suite##_##name##_Body);          \
    static void suite##_##name##_Body()

#define UNGAR_GTEST_NONFATAL_(cond, what) ::testing::Message(!(cond), __FILE__, __LINE__, what)
// fatal: on failure report (with any streamed operands) and return from the test body
#define UNGAR_GTEST_FATAL_(cond, what) \
    if (cond)                          \
        ;                              \
    else                               \
        return ::testing::Voidify() & ::testing::Message(true, __FILE__, __LINE__, what)

#define EXPECT_TRUE(c) UNGAR_GTEST_NONFATAL_(static_cast<bool>(c), "Value of: " #c "\n  Actual: false\nExpected: true ")
#define EXPECT_FALSE(c) UNGAR_GTEST_NONFATAL_(!static_cast<bool>(c), "Value of: " #c "\n  Actual: true\nExpected: false ")
#define EXPECT_EQ(a, b) UNGAR_GTEST_NONFATAL_((a) == (b), "Expected equality of: " #a " and " #b " ")
#define EXPECT_NE(a, b) UNGAR_GTEST_NONFATAL_((a) != (b), "Expected: (" #a ") != (" #b ") ")
#define EXPECT_LT(a, b) UNGAR_GTEST_NONFATAL_((a) < (b), "Expected: (" #a ") < (" #b ") ")
#define EXPECT_LE(a, b) UNGAR_GTEST_NONFATAL_((a) <= (b), "Expected: (" #a ") <= (" #b ") ")
#define EXPECT_GT(a, b) UNGAR_GTEST_NONFATAL_((a) > (b), "Expected: (" #a ") > (" #b ") ")
#define EXPECT_GE(a, b) UNGAR_GTEST_NONFATAL_((a) >= (b), "Expected: (" #a ") >= (" #b ") ")
#define EXPECT_NEAR(a, b, tol) UNGAR_GTEST_NONFATAL_(std::abs((a) - (b)) <= (tol), "The difference between " #a " and " #b " exceeds " #tol " ")
#define EXPECT_DOUBLE_EQ(a, b) UNGAR_GTEST_NONFATAL_(std::abs((a) - (b)) <= 4 * 2.220446049250313e-16 * std::max(std::abs(a), std::abs(b)), "Expected equality of: " #a " and " #b " ")

#define EXPECT_PRED2(pred, a, b) UNGAR_GTEST_NONFATAL_((pred)((a), (b)), #pred "(" #a ", " #b ") evaluates to false ")
#define ASSERT_PRED2(pred, a, b) UNGAR_GTEST_FATAL_((pred)((a), (b)), #pred "(" #a ", " #b ") evaluates to false ")
#define ASSERT_TRUE(c) UNGAR_GTEST_FATAL_(static_cast<bool>(c), "Value of: " #c "\n  Actual: false\nExpected: true ")
#define ASSERT_FALSE(c) UNGAR_GTEST_FATAL_(!static_cast<bool>(c), "Value of: " #c "\n  Actual: true\nExpected: false ")
#define ASSERT_EQ(a, b) UNGAR_GTEST_FATAL_((a) == (b), "Expected equality of: " #a " and " #b " ")
#define ASSERT_NE(a, b) UNGAR_GTEST_FATAL_((a) != (b), "Expected: (" #a ") != (" #b ") ")
#define ASSERT_LT(a, b) UNGAR_GTEST_FATAL_((a) < (b), "Expected: (" #a ") < (" #b ") ")
#define ASSERT_LE(a, b) UNGAR_GTEST_FATAL_((a) <= (b), "Expected: (" #a ") <= (" #b ") ")
#define ASSERT_GT(a, b) UNGAR_GTEST_FATAL_((a) > (b), "Expected: (" #a ") > (" #b ") ")
#define ASSERT_GE(a, b) UNGAR_GTEST_FATAL_((a) >= (b), "Expected: (" #a ") >= (" #b ") ")
#define ASSERT_NEAR(a, b, tol) UNGAR_GTEST_FATAL_(std::abs((a) - (b)) <= (tol), "The difference between " #a " and " #b " exceeds " #tol " ")
