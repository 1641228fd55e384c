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
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace {

// A file's bytes as Python's text mode reads them (universal newlines: "\r\n" and a lone "\r" both read as "\n").
// The file is mapped, not copied; only a file that does contain '\r' is rewritten into a private copy.
class FileText {
  public:
    FileText() = default;
    FileText(const FileText &) = delete;
    FileText &operator=(const FileText &) = delete;
    ~FileText() {
        if (map_) ::munmap(map_, len_);
    }

    bool open(const char *path, char *err, int errlen) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) {
            std::snprintf(err, errlen, "cannot open %s", path);
            return false;
        }
        struct stat st;
        if (::fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {
            ::close(fd);
            std::snprintf(err, errlen, "cannot read %s", path);
            return false;
        }
        len_ = size_t(st.st_size);
        if (len_ > 0) {
            void *m = ::mmap(nullptr, len_, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m == MAP_FAILED) {
                ::close(fd);
                std::snprintf(err, errlen, "cannot map %s", path);
                return false;
            }
            map_ = m;
        }
        ::close(fd);
        text_ = len_ ? std::string_view(static_cast<const char *>(map_), len_) : std::string_view("", 0);
        if (len_ == 0 || std::memchr(map_, '\r', len_) == nullptr) return true;
        copy_.resize(len_);
        size_t w = 0;
        for (size_t r = 0; r < len_; ++r) {
            if (text_[r] == '\r') {
                copy_[w++] = '\n';
                if (r + 1 < len_ && text_[r + 1] == '\n') ++r;
            } else {
                copy_[w++] = text_[r];
            }
        }
        copy_.resize(w);
        text_ = copy_;
        return true;
    }

    std::string_view text() const { return text_; }

  private:
    void *map_ = nullptr;
    size_t len_ = 0;
    std::string copy_;
    std::string_view text_;
};

// str.strip() without arguments removes every character with str.isspace(): the ASCII ones (\t \n \v \f \r,
// 0x1c-0x1f, space) and, in UTF-8, U+0085, U+00A0, U+1680, U+2000-U+200A, U+2028, U+2029, U+202F, U+205F, U+3000.
// Length in bytes of the whitespace character that STARTS at s[0] / ENDS at s[n-1], or 0.
inline bool is_ascii_space(unsigned char c) { return (c >= 9 && c <= 13) || (c >= 0x1c && c <= 0x20); }

inline bool is_space_2(unsigned char a, unsigned char b) {            // U+0085, U+00A0
    return a == 0xc2 && (b == 0x85 || b == 0xa0);
}

inline bool is_space_3(unsigned char a, unsigned char b, unsigned char c) {
    if (a == 0xe1) return b == 0x9a && c == 0x80;                                              // U+1680
    if (a == 0xe2) {
        if (b == 0x80) return (c >= 0x80 && c <= 0x8a) || c == 0xa8 || c == 0xa9 || c == 0xaf;  // U+2000-200A, 2028, 2029, 202F
        return b == 0x81 && c == 0x9f;                                                          // U+205F
    }
    return a == 0xe3 && b == 0x80 && c == 0x80;                                                // U+3000
}

inline size_t space_at_front(std::string_view s) {
    const auto u = [&](size_t i) { return static_cast<unsigned char>(s[i]); };
    if (s.empty()) return 0;
    if (is_ascii_space(u(0))) return 1;
    if (s.size() >= 2 && is_space_2(u(0), u(1))) return 2;
    if (s.size() >= 3 && is_space_3(u(0), u(1), u(2))) return 3;
    return 0;
}

inline size_t space_at_back(std::string_view s) {
    const size_t n = s.size();
    const auto u = [&](size_t i) { return static_cast<unsigned char>(s[i]); };
    if (n == 0) return 0;
    if (is_ascii_space(u(n - 1))) return 1;
    if (n >= 2 && is_space_2(u(n - 2), u(n - 1))) return 2;
    if (n >= 3 && is_space_3(u(n - 3), u(n - 2), u(n - 1))) return 3;
    return 0;
}

std::string_view stripped(std::string_view s) {
    for (size_t k; (k = space_at_front(s)) != 0;) s.remove_prefix(k);
    for (size_t k; (k = space_at_back(s)) != 0;) s.remove_suffix(k);
    return s;
}

// Is [p, p + n) well-formed UTF-8 (Python's strict decoder: no overlong forms, no surrogates, nothing above U+10FFFF)?
// The files are read in text mode upstream, so anything else is a UnicodeDecodeError there.
bool valid_utf8(const char *ptr, size_t n) {
    const unsigned char *p = reinterpret_cast<const unsigned char *>(ptr), *end = p + n;
    while (p < end) {
        if (end - p >= 8) {                                     // eight ASCII bytes at a time
            uint64_t w;
            std::memcpy(&w, p, 8);
            if ((w & 0x8080808080808080ull) == 0) {
                p += 8;
                continue;
            }
        }
        const unsigned char c = *p;
        if (c < 0x80) {
            ++p;
        } else if (c >= 0xc2 && c <= 0xdf) {
            if (end - p < 2 || (p[1] & 0xc0) != 0x80) return false;
            p += 2;
        } else if (c >= 0xe0 && c <= 0xef) {
            if (end - p < 3 || (p[1] & 0xc0) != 0x80 || (p[2] & 0xc0) != 0x80) return false;
            if ((c == 0xe0 && p[1] < 0xa0) || (c == 0xed && p[1] > 0x9f)) return false;      // overlong / surrogate
            p += 3;
        } else if (c >= 0xf0 && c <= 0xf4) {
            if (end - p < 4 || (p[1] & 0xc0) != 0x80 || (p[2] & 0xc0) != 0x80 || (p[3] & 0xc0) != 0x80) return false;
            if ((c == 0xf0 && p[1] < 0x90) || (c == 0xf4 && p[1] > 0x8f)) return false;      // overlong / beyond U+10FFFF
            p += 4;
        } else {
            return false;
        }
    }
    return true;
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

// id -> index of its FIRST line in V.  Open addressing over a power-of-two table at most half full; a slot keeps the
// 64-bit hash (compared before the bytes), so a lookup is one cache miss in the common case -- and the parser hides
// those behind each other by hashing and prefetching a batch of lines before it resolves them (std::unordered_map:
// ~400 ns of thread time per edge at |V| = 2M; this: ~60).
class IdTable {
  public:
    static uint64_t hash(const char *p, size_t n) {
        auto mix = [](uint64_t a, uint64_t b) {
            const __uint128_t m = (__uint128_t)a * b;
            return uint64_t(m) ^ uint64_t(m >> 64);
        };
        uint64_t h = 0x9e3779b97f4a7c15ull ^ (uint64_t(n) * 0xff51afd7ed558ccdull);
        while (n >= 8) {
            uint64_t w;
            std::memcpy(&w, p, 8);
            h = mix(h ^ w, 0xc4ceb9fe1a85ec53ull);
            p += 8;
            n -= 8;
        }
        uint64_t w = 0;
        std::memcpy(&w, p, n);
        return mix(h ^ w, 0x2545f4914f6cdd1dull);
    }

    void build(std::string_view body) {
        size_t lines = 1;
        for (char c : body) lines += c == '\n';
        size_t cap = 16;
        while (cap < 2 * lines) cap <<= 1;
        slots_.assign(cap, Slot{0, nullptr, 0, -1});
        mask_ = cap - 1;
        // ids in batches, hashed and prefetched before they are placed (the table is far larger than the caches)
        constexpr int kBatch = 16;
        std::string_view ids[kBatch];
        uint64_t hs[kBatch];
        int held = 0;
        int64_t i = 0;
        auto place = [&]() {
            for (int b = 0; b < held; ++b, ++i) {
                const std::string_view id = ids[b];
                for (size_t at = size_t(hs[b]) & mask_;; at = (at + 1) & mask_) {
                    Slot &s = slots_[at];
                    if (s.idx < 0) {
                        s = Slot{hs[b], id.data(), uint32_t(id.size()), i};
                        break;
                    }
                    if (s.h == hs[b] && s.len == id.size() && std::memcmp(s.p, id.data(), id.size()) == 0) break;  // keep the first
                }
            }
            held = 0;
        };
        for_each_line(body, [&](std::string_view id) {
            ids[held] = id;
            hs[held] = hash(id.data(), id.size());
            prefetch(hs[held]);
            if (++held == kBatch) place();
        });
        place();
    }

    void prefetch(uint64_t h) const { __builtin_prefetch(&slots_[size_t(h) & mask_]); }

    int64_t find(uint64_t h, std::string_view id) const {
        for (size_t at = size_t(h) & mask_;; at = (at + 1) & mask_) {
            const Slot &s = slots_[at];
            if (s.idx < 0) return -1;
            if (s.h == h && s.len == id.size() && std::memcmp(s.p, id.data(), id.size()) == 0) return s.idx;
        }
    }

  private:
    struct Slot {
        uint64_t h;
        const char *p;
        uint32_t len;
        int64_t idx;  // -1: empty
    };
    std::vector<Slot> slots_;
    size_t mask_ = 0;
};


}  // namespace

extern "C" {

// Number of '\n'-separated fields of the stripped file (what len(read().strip().split('\n')) gives); -1 on error.
int64_t clane_count_lines(const char *path, char *err, int errlen) {
    FileText file;
    if (!file.open(path, err, errlen)) return -1;
    const std::string_view body = stripped(file.text());
    const auto start = piece_starts(body, worker_count(body.size()));
    int64_t n = 0;
    for (int64_t c : count_fields(body, start)) n += c;
    return n;
}

// src[k], dst[k] <- vertex indices of line k of E.  Returns the number of edges, or -1 (file error),
// -2 (malformed line), -3 (unknown vertex id), -4 (a file is not valid UTF-8: text mode upstream raises UnicodeDecodeError); err holds the message of the FIRST offending line.
// The E file is cut at line boundaries and parsed by up to 16 threads (the id -> index map is read-only by then).
int64_t clane_parse_edges(const char *v_path, const char *e_path, int64_t *src, int64_t *dst, int64_t capacity,
                          char *err, int errlen) {
    FileText v_file, e_file;
    if (!v_file.open(v_path, err, errlen) || !e_file.open(e_path, err, errlen)) return -1;
    if (!valid_utf8(v_file.text().data(), v_file.text().size())) {
        std::snprintf(err, errlen, "V is not valid UTF-8");
        return -4;
    }
    IdTable first;
    first.build(stripped(v_file.text()));
    const std::string_view body = stripped(e_file.text());
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
        if (!valid_utf8(part.data(), part.size())) {         // pieces end at line ends: no character straddles two
            fail = Fail{k, -4, "E is not valid UTF-8"};
            return;
        }
        // lines in batches: cut + hash + prefetch the whole batch, then resolve it (the table misses overlap)
        constexpr int kBatch = 16;
        struct Pending {
            std::string_view line;
            size_t tab;
            uint64_t hs, hd;
        };
        Pending batch[kBatch];
        int held = 0;
        auto resolve = [&]() {
            for (int i = 0; i < held && !fail.status; ++i) {
                const Pending &q = batch[i];
                char buf[160];
                if (q.tab == std::string_view::npos) {
                    std::snprintf(buf, sizeof buf, "E line %lld: expected 'src\\tdst', got '%.*s'", (long long)(k + 1),
                                  int(q.line.size() < 80 ? q.line.size() : 80), q.line.data());
                    fail = Fail{k, -2, buf};
                    return;
                }
                const std::string_view a = q.line.substr(0, q.tab), b = q.line.substr(q.tab + 1);
                const int64_t si = first.find(q.hs, a), di = first.find(q.hd, b);
                if (si < 0 || di < 0) {
                    const std::string_view bad = si < 0 ? a : b;
                    std::snprintf(buf, sizeof buf, "'%.*s' is not in list", int(bad.size() < 80 ? bad.size() : 80), bad.data());
                    fail = Fail{k, -3, buf};
                    return;
                }
                src[k] = si;
                dst[k] = di;
                ++k;
            }
            held = 0;
        };
        for_each_line(part, [&](std::string_view line) {
            if (fail.status) return;
            Pending &q = batch[held];
            q.line = line;
            q.tab = line.find('\t');
            if (q.tab != std::string_view::npos && line.find('\t', q.tab + 1) != std::string_view::npos)
                q.tab = std::string_view::npos;                  // more than one tab: malformed, like no tab
            if (q.tab != std::string_view::npos) {
                q.hs = IdTable::hash(line.data(), q.tab);
                q.hd = IdTable::hash(line.data() + q.tab + 1, line.size() - q.tab - 1);
                first.prefetch(q.hs);
                first.prefetch(q.hd);
            }
            if (++held == kBatch) resolve();
        });
        resolve();
    });
    for (const Fail &f : fails) {                              // upstream decodes the whole file before it parses a line
        if (f.status == -4) {
            std::snprintf(err, errlen, "%s", f.msg.c_str());
            return -4;
        }
    }
    for (const Fail &f : fails) {                              // pieces are in file order: the first failure wins
        if (f.status) {
            std::snprintf(err, errlen, "%s", f.msg.c_str());
            return f.status;
        }
    }
    return off.back();
}

}  // extern "C"
