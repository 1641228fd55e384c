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
#include <string>
#include <string_view>
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

}  // namespace

extern "C" {

// Number of '\n'-separated fields of the stripped file (what len(read().strip().split('\n')) gives); -1 on error.
int64_t clane_count_lines(const char *path, char *err, int errlen) {
    std::string buf;
    if (!read_file(path, buf, err, errlen)) return -1;
    int64_t n = 0;
    for_each_line(stripped(buf), [&](std::string_view) { ++n; });
    return n;
}

// src[k], dst[k] <- vertex indices of line k of E.  Returns the number of edges, or -1 (file error),
// -2 (malformed line), -3 (unknown vertex id); err holds the message.
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
    int64_t k = 0, status = 0;
    for_each_line(stripped(ebuf), [&](std::string_view line) {
        if (status) return;
        const size_t tab = line.find('\t');
        if (tab == std::string_view::npos || line.find('\t', tab + 1) != std::string_view::npos) {
            std::snprintf(err, errlen, "E line %lld: expected 'src\\tdst', got '%.*s'", (long long)(k + 1),
                          int(line.size() < 80 ? line.size() : 80), line.data());
            status = -2;
            return;
        }
        const auto s = first.find(line.substr(0, tab)), d = first.find(line.substr(tab + 1));
        if (s == first.end() || d == first.end()) {
            const std::string_view bad = s == first.end() ? line.substr(0, tab) : line.substr(tab + 1);
            std::snprintf(err, errlen, "'%.*s' is not in list", int(bad.size() < 80 ? bad.size() : 80), bad.data());
            status = -3;
            return;
        }
        if (k >= capacity) {
            std::snprintf(err, errlen, "output buffers too small (%lld)", (long long)capacity);
            status = -1;
            return;
        }
        src[k] = s->second;
        dst[k] = d->second;
        ++k;
    });
    return status ? status : k;
}

}  // extern "C"
