// Driver for tests/sanitize_host.sh: runs the host-side native code (csrc/host_loader.cpp and the C oracle) under
// AddressSanitizer + UBSan on the CPU.  Not part of the product.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

extern "C" {
int64_t clane_count_lines(const char *path, char *err, int errlen);
int64_t clane_parse_edges(const char *v_path, const char *e_path, int64_t *src, int64_t *dst, int64_t capacity,
                          char *err, int errlen);
double clane_c_sweep_f32(const int64_t *rowptr, const int32_t *colidx, const float *P, int64_t V, int32_t d,
                         const float *X, const float *Zold, double gamma, float *Znew);
double clane_c_build_P_f32(const int64_t *rowptr, const int32_t *colidx, int64_t V, int32_t d, const float *Z, float *P);
void clane_c_set_threads(int n);
}

static void write_file(const std::string &path, const std::string &body) {
    FILE *f = std::fopen(path.c_str(), "wb");
    if (!f) std::abort();
    std::fwrite(body.data(), 1, body.size(), f);
    std::fclose(f);
}

static int64_t parse(const std::string &dir, const std::string &v, const std::string &e, std::vector<int64_t> &src,
                     std::vector<int64_t> &dst, std::string &msg) {
    write_file(dir + "/V", v);
    write_file(dir + "/E", e);
    char err[256] = {0};
    const int64_t n = clane_count_lines((dir + "/E").c_str(), err, sizeof err);
    if (n < 0) {
        msg = err;
        return n;
    }
    src.assign(size_t(n), -1);
    dst.assign(size_t(n), -1);
    const int64_t got = clane_parse_edges((dir + "/V").c_str(), (dir + "/E").c_str(), src.data(), dst.data(), n, err, sizeof err);
    msg = err;
    return got;
}

#define CHECK(c)                                                     \
    do {                                                             \
        if (!(c)) {                                                  \
            std::fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); \
            return 1;                                                \
        }                                                            \
    } while (0)

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const std::string dir = argv[1];
    std::vector<int64_t> s, d;
    std::string msg;
    CHECK(parse(dir, "a\nb\nc\n", "a\tb\nb\tc\nc\ta\n", s, d, msg) == 3 && s[2] == 2 && d[2] == 0);
    CHECK(parse(dir, "\n\na\r\nb\rc  \n", "a\tb\r\nc\ta", s, d, msg) == 2 && s[1] == 2);      // CR / CRLF, outer whitespace
    CHECK(parse(dir, "a\nb", "a b", s, d, msg) == -2);                                         // no tab
    CHECK(parse(dir, "a\nb", "a\tb\tc", s, d, msg) == -2);                                     // two tabs
    CHECK(parse(dir, "a\nb", "a\tz", s, d, msg) == -3 && msg.find("'z'") != std::string::npos);
    CHECK(parse(dir, "a\na\nb", "a\tb", s, d, msg) == 1 && s[0] == 0);                          // first occurrence wins
    CHECK(parse(dir, "a", "", s, d, msg) == -2);                                               // empty E: one empty line
    CHECK(parse(dir, "\x1c\xc2\xa0 a\nb\xe3\x80\x80\xe2\x80\xa8", "\xc2\x85" "a\tb\xe2\x81\x9f\x1f", s, d, msg) == 1 && s[0] == 0 && d[0] == 1);  // Unicode spaces around the files
    CHECK(parse(dir, "a\xc2\xa0" "b\nc", "a\xc2\xa0" "b\tc", s, d, msg) == 1 && s[0] == 0 && d[0] == 1);   // ... and inside an id
    CHECK(parse(dir, "a\nb", "a\tb\nb\t\xff", s, d, msg) == -4);                                // not UTF-8: upstream's text mode raises
    CHECK(parse(dir, "a\n\xed\xa0\x80", "a\ta", s, d, msg) == -4);                              // a surrogate in V
    CHECK(parse(dir, "a\nb", "zz\tb\nb\t\xc0\xaf", s, d, msg) == -4);                          // decoding fails before any line is parsed
    CHECK(parse(dir, "\xf0\x9f\x98\x80\nb", "\xf0\x9f\x98\x80\tb", s, d, msg) == 1 && s[0] == 0);  // 4-byte characters are fine
    CHECK(parse(dir, "", "x\ty", s, d, msg) == -3);                                            // empty V file: one vertex named ""
    CHECK(parse(dir, "", "\t", s, d, msg) == -2);                                              // E strips to one empty line
    CHECK(parse(dir, "a\n\nb", "a\t\n\tb", s, d, msg) == 2 && d[0] == 1 && s[1] == 1 && d[1] == 2);   // the empty id
    CHECK(parse(dir, "a b\nc", "a b\tc", s, d, msg) == 1 && s[0] == 0 && d[0] == 1);            // a space is part of an id
    {                                                                                          // ids of 1..40 bytes (hash: 8-byte words + tail)
        std::string v, e;
        const int n = 41;
        for (int i = 0; i < n; ++i) v += std::string(size_t(i + 1), char('a' + i % 26)) + "\n";
        v += "last";
        for (int i = 0; i < n; ++i) e += std::string(size_t(i + 1), char('a' + i % 26)) + "\tlast\n";
        CHECK(parse(dir, v, e, s, d, msg) == n);
        for (int i = 0; i < n; ++i) CHECK(s[size_t(i)] == i && d[size_t(i)] == n);
    }
    {                                                                                          // big enough for 16 threads
        std::string v, e;
        const int n = 200000;
        for (int i = 0; i < n; ++i) v += "v" + std::to_string(i) + "\n";
        for (int i = 0; i < 3 * n; ++i)
            e += "v" + std::to_string((int64_t(i) * 7919) % n) + "\tv" + std::to_string((int64_t(i) * 104729 + 1) % n) + "\n";
        CHECK(parse(dir, v, e, s, d, msg) == 3 * n);
        for (int i = 0; i < 3 * n; i += 997) CHECK(s[size_t(i)] == (int64_t(i) * 7919) % n && d[size_t(i)] == (int64_t(i) * 104729 + 1) % n);
        e += "v1\tnobody\n";
        CHECK(parse(dir, v, e, s, d, msg) == -3);
    }
    if (argc < 3) {   // the C oracle on a small graph (skipped in the ThreadSanitizer build: libgomp is not instrumented) with an empty row, a self-loop and a hub
        const int64_t V = 6;
        const int32_t dd = 5;
        const int64_t rowptr[] = {0, 2, 2, 3, 8, 9, 10};
        const int32_t colidx[] = {1, 2, 2, 0, 1, 2, 3, 5, 0, 4};
        std::vector<float> X(size_t(V * dd)), Z(size_t(V * dd)), P(10);
        for (size_t i = 0; i < X.size(); ++i) X[i] = float((i * 37) % 11) - 5.0f;
        for (int t = 1; t <= 3; ++t) {
            clane_c_set_threads(t);
            const double D = clane_c_build_P_f32(rowptr, colidx, V, dd, X.data(), P.data());
            CHECK(D > 0);
            float rs = P[3] + P[4] + P[5] + P[6] + P[7];
            CHECK(rs > 0.9999f && rs < 1.0001f);
            const double delta = clane_c_sweep_f32(rowptr, colidx, P.data(), V, dd, X.data(), X.data(), 0.76, Z.data());
            CHECK(delta > 0);
            for (int k = 0; k < dd; ++k) CHECK(Z[size_t(1 * dd + k)] == X[size_t(1 * dd + k)]);   // sink row keeps x
        }
    }
    std::puts("sanitize_host: ok");
    return 0;
}
