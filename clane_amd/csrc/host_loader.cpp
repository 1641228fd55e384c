// libclane_host.so -- native O(|V| + |E|) parser for the reference's graph files (host CPU, no GPU).
//
// Replaces the loader of clane/graph.py:43-47,72-89, which resolves each edge endpoint with
// `vertex_ids.index(id)` -- O(|V|) per endpoint, days at |V| = 2M.  Same semantics:
//   V : whole file stripped of leading/trailing whitespace, split on '\n'; an id is the raw line.
//   E : same; every line must split on '\t' into exactly two ids; an id resolves to the index of its
//       FIRST occurrence in V; anything else is an error (the reference raises ValueError).
// C ABI: the caller passes output buffers it owns (two int64 arrays of clane_count_lines(E) entries).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

bool read_file(const char *path, std::string &out, char *err, int errlen) {
    FILE *f = std::fopen(path, "rb");
    if (!f) {
        std::snprintf(err, errlen, "cannot open %s", path);
        return false;
    }
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out.resize(size_t(n));
    const size_t got = n ? std::fread(out.data(), 1, size_t(n), f) : 0;
    std::fclose(f);
    if (got != size_t(n)) {
        std::snprintf(err, errlen, "short read on %s", path);
        return false;
    }
    // Python opens the files in text mode: universal newlines, "\r\n" and a lone "\r" both read as "\n".
    if (std::memchr(out.data(), '\r', out.size()) == nullptr) return true;
    size_t w = 0;
    for (size_t r = 0; r < out.size(); ++r) {
        if (out[r] == '\r') {
            out[w++] = '\n';
            if (r + 1 < out.size() && out[r + 1] == '\n') ++r;
        } else {
            out[w++] = out[r];
        }
    }
    out.resize(w);
    return true;
}

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

std::string_view stripped(const std::string &s) {
    size_t a = 0, b = s.size();
    while (a < b && is_space(s[a])) ++a;
    while (b > a && is_space(s[b - 1])) --b;
    return std::string_view(s).substr(a, b - a);
}

template <typename F>
void for_each_line(std::string_view body, F &&f) {  // Python's str.split('\n'): n separators -> n+1 fields
    size_t pos = 0;
    for (;;) {
        const size_t nl = body.find('\n', pos);
        if (nl == std::string_view::npos) {
            f(body.substr(pos));
            return;
        }
        f(body.substr(pos, nl - pos));
        pos = nl + 1;
    }
}

// Cut `body` into pieces that start right after a '\n' (piece 0 at 0), about equal in bytes.
std::vector<size_t> piece_starts(std::string_view body, int pieces) {
    std::vector<size_t> start{0};
    for (int t = 1; t < pieces; ++t) {
        size_t p = body.size() / size_t(pieces) * size_t(t);
        if (p <= start.back()) continue;
        const size_t nl = body.find('\n', p);
        if (nl == std::string_view::npos) break;
        if (nl + 1 > start.back()) start.push_back(nl + 1);
    }
    start.push_back(body.size());
    return start;
}

int worker_count(size_t bytes) {
    if (bytes < (size_t(1) << 22)) return 1;
    const unsigned hw = std::thread::hardware_concurrency();
    return int(std::min<unsigned>(hw ? hw : 1, 16));
}

template <typename F>
void run_pieces(int n, F &&f) {
    if (n == 1) {
        f(0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 0; t < n; ++t) th.emplace_back([&f, t] { f(t); });
    for (auto &x : th) x.join();
}

// Fields per piece under Python's split('\n') (n separators -> n+1 fields: the last piece owns the +1).
std::vector<int64_t> count_fields(std::string_view body, const std::vector<size_t> &start) {
    const int n = int(start.size()) - 1;
    std::vector<int64_t> cnt(size_t(n), 0);
    run_pieces(n, [&](int t) {
        const char *p = body.data() + start[t], *end = body.data() + start[t + 1];
        int64_t c = 0;
        while (p < end) {
            const char *q = static_cast<const char *>(std::memchr(p, '\n', size_t(end - p)));
            if (!q) break;
            ++c;
            p = q + 1;
        }
        cnt[size_t(t)] = c + (t == n - 1 ? 1 : 0);
    });
    return cnt;
}

}  // namespace

extern "C" {

// Number of '\n'-separated fields of the stripped file (what len(read().strip().split('\n')) gives); -1 on error.
int64_t clane_count_lines(const char *path, char *err, int errlen) {
    std::string buf;
    if (!read_file(path, buf, err, errlen)) return -1;
    const std::string_view body = stripped(buf);
    const auto start = piece_starts(body, worker_count(body.size()));
    int64_t n = 0;
    for (int64_t c : count_fields(body, start)) n += c;
    return n;
}

// src[k], dst[k] <- vertex indices of line k of E.  Returns the number of edges, or -1 (file error),
// -2 (malformed line), -3 (unknown vertex id); err holds the message of the FIRST offending line.
// The E file is cut at line boundaries and parsed by up to 16 threads (the id -> index map is read-only by then).
int64_t clane_parse_edges(const char *v_path, const char *e_path, int64_t *src, int64_t *dst, int64_t capacity,
                          char *err, int errlen) {
    std::string vbuf, ebuf;
    if (!read_file(v_path, vbuf, err, errlen) || !read_file(e_path, ebuf, err, errlen)) return -1;
    std::unordered_map<std::string_view, int64_t> first;
    {
        int64_t n = 0;
        for_each_line(stripped(vbuf), [&](std::string_view) { ++n; });
        first.reserve(size_t(n) * 2);
        int64_t i = 0;
        for_each_line(stripped(vbuf), [&](std::string_view id) { first.emplace(id, i++); });  // emplace keeps the first
    }
    const std::string_view body = stripped(ebuf);
    const auto start = piece_starts(body, worker_count(body.size()));
    const int pieces = int(start.size()) - 1;
    const auto cnt = count_fields(body, start);
    std::vector<int64_t> off(size_t(pieces) + 1, 0);
    for (int t = 0; t < pieces; ++t) off[size_t(t) + 1] = off[size_t(t)] + cnt[size_t(t)];
    if (off.back() > capacity) {
        std::snprintf(err, errlen, "output buffers too small (%lld)", (long long)capacity);
        return -1;
    }
    struct Fail {
        int64_t line = -1;
        int status = 0;
        std::string msg;
    };
    std::vector<Fail> fails;
    fails.resize(size_t(pieces));
    run_pieces(pieces, [&](int t) {
        int64_t k = off[size_t(t)];
        Fail &fail = fails[size_t(t)];
        std::string_view part = body.substr(start[t], start[t + 1] - start[t]);
        if (t < pieces - 1) part.remove_suffix(1);           // the '\n' that ends the piece's last line
        for_each_line(part, [&](std::string_view line) {
            if (fail.status) return;
            char buf[160];
            const size_t tab = line.find('\t');
            if (tab == std::string_view::npos || line.find('\t', tab + 1) != std::string_view::npos) {
                std::snprintf(buf, sizeof buf, "E line %lld: expected 'src\\tdst', got '%.*s'", (long long)(k + 1),
                              int(line.size() < 80 ? line.size() : 80), line.data());
                fail = Fail{k, -2, buf};
                return;
            }
            const auto s = first.find(line.substr(0, tab)), d = first.find(line.substr(tab + 1));
            if (s == first.end() || d == first.end()) {
                const std::string_view bad = s == first.end() ? line.substr(0, tab) : line.substr(tab + 1);
                std::snprintf(buf, sizeof buf, "'%.*s' is not in list", int(bad.size() < 80 ? bad.size() : 80), bad.data());
                fail = Fail{k, -3, buf};
                return;
            }
            src[k] = s->second;
            dst[k] = d->second;
            ++k;
        });
    });
    for (const Fail &f : fails) {                              // pieces are in file order: the first failure wins
        if (f.status) {
            std::snprintf(err, errlen, "%s", f.msg.c_str());
            return f.status;
        }
    }
    return off.back();
}

}  // extern "C"
