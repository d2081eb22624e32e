// Wavefront OBJ ingest on the host (no device code): what obj_to_actor gets out of vtkOBJReader
// (reference src/mvlm/utils/utils3d.py:16-24) - float32 points, one texture coordinate per point
// (a point is duplicated when it is used with several `vt` indices), polygons as triangle fans,
// `.mtl` / normals / groups ignored.  A 100k-triangle scan is ~6 MB of text; this reader is what
// keeps file ingest (tens of ms) below the GPU time of the views it feeds.
//
// The corner numbering is the contract with mvlm_amd/utils/mesh_io.py:_parse_obj (the pure-Python
// statement of the same rules, kept as the cross-check in tests): corners are numbered in order of
// first use of each (v, vt) pair, negative indices are relative to the elements read so far.
#include <sched.h>

#include <atomic>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mvlm_hip.h"
#include "mesh_obj.h"

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\v' || c == '\f'; }

// Decimal -> double, correctly rounded.  Fast path (Clinger): up to 19 digits collected exactly in a
// 64-bit mantissa below 2^53 and a power of ten no larger than 10^22 are both exact doubles, so one
// IEEE multiply/divide rounds once.  Anything else goes through strtod.
bool parse_double(const char* s, const char* end, double* out) {
    static const double P10[] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    const char* p = s;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) neg = *p++ == '-';
    uint64_t mant = 0;
    int digits = 0, frac = 0;
    bool any = false, simple = true;
    while (p < end && *p >= '0' && *p <= '9') {
        if (digits < 19) { mant = mant * 10 + uint64_t(*p - '0'); if (mant) ++digits; } else simple = false;
        ++p; any = true;
    }
    if (p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') {
            if (digits < 19) { mant = mant * 10 + uint64_t(*p - '0'); if (mant) ++digits; ++frac; } else simple = false;
            ++p; any = true;
        }
    }
    int ex = 0;
    if (any && p < end && (*p == 'e' || *p == 'E')) {
        const char* q = p + 1;
        bool eneg = false;
        if (q < end && (*q == '-' || *q == '+')) eneg = *q++ == '-';
        if (q < end && *q >= '0' && *q <= '9') {
            while (q < end && *q >= '0' && *q <= '9') { if (ex < 10000) ex = ex * 10 + (*q - '0'); ++q; }
            if (eneg) ex = -ex;
            p = q;
        }
    }
    if (any && p == end && simple && mant < (1ull << 53)) {
        const int e10 = ex - frac;
        if (e10 >= -22 && e10 <= 22) {
            double v = double(mant);
            v = e10 < 0 ? v / P10[-e10] : v * P10[e10];
            *out = neg ? -v : v;
            return true;
        }
    }
    // slow path: inf / nan / long mantissas / large exponents
    char buf[128];
    const size_t n = size_t(end - s);
    if (n == 0 || n >= sizeof buf) return false;
    memcpy(buf, s, n);
    buf[n] = 0;
    if ((buf[0] == '0' || ((buf[0] == '-' || buf[0] == '+') && buf[1] == '0')) && strpbrk(buf, "xX")) return false;  // no hex floats
    char* stop = nullptr;
    errno = 0;
    const double v = strtod(buf, &stop);
    if (stop != buf + n) return false;
    *out = v;
    return true;
}

bool parse_index(const char* s, const char* end, long long* out) {
    const char* p = s;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) neg = *p++ == '-';
    if (p == end) return false;
    long long v = 0;
    for (; p < end; ++p) {
        if (*p < '0' || *p > '9') return false;
        if (v < (1ll << 40)) v = v * 10 + (*p - '0');
    }
    *out = neg ? -v : v;
    return true;
}

struct CornerMap {  // (v, vt) -> corner id, open addressing
    std::vector<uint64_t> keys;
    std::vector<int32_t> vals;
    size_t used = 0, mask = 0;
    static constexpr uint64_t EMPTY = ~0ull;
    CornerMap() { rehash(1u << 16); }
    static uint64_t mix(uint64_t k) {
        k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
        return k;
    }
    void rehash(size_t cap) {
        std::vector<uint64_t> ok(cap, EMPTY);
        std::vector<int32_t> ov(cap, 0);
        ok.swap(keys); ov.swap(vals);
        mask = cap - 1;
        for (size_t i = 0; i < ok.size(); ++i)
            if (ok[i] != EMPTY) {
                size_t h = mix(ok[i]) & mask;
                while (keys[h] != EMPTY) h = (h + 1) & mask;
                keys[h] = ok[i]; vals[h] = ov[i];
            }
    }
    // returns the id for key, inserting `fresh` when absent
    int32_t get(uint64_t key, int32_t fresh, bool* inserted) {
        if ((used + 1) * 2 > keys.size()) rehash(keys.size() * 2);
        size_t h = mix(key) & mask;
        while (keys[h] != EMPTY) {
            if (keys[h] == key) { *inserted = false; return vals[h]; }
            h = (h + 1) & mask;
        }
        keys[h] = key; vals[h] = fresh; ++used; *inserted = true;
        return fresh;
    }
};

void set_err(char* err, int err_len, const std::string& msg) {
    if (err && err_len > 0) snprintf(err, size_t(err_len), "%s", msg.c_str());
}

// ---- chunk-parallel parse --------------------------------------------------------------------------------------
// The text is cut at line ends into one chunk per thread.  A chunk is parsed on its own: positions and texture
// coordinates into local arrays, faces as raw index pairs together with the number of elements the chunk had seen when
// the face came (OBJ's negative indices count back from there).  What needs the whole file - the element bases of a
// chunk (prefix sum of the counts), the numbering of the (v, vt) corners in order of first use, the first error in file
// order - follows in one sequential pass over the already parsed faces.  Results are those of a single-threaded parse,
// byte for byte, for any number of chunks (tests/test_host_logic.py compares 1, 2, 3, 7 threads and the Python statement).
constexpr long long NO_TEX = -(1ll << 62);

struct Face {
    uint32_t first, count;          // corners [first, first + count) of the chunk's corner arrays
    uint32_t pos_before, tex_before;  // elements this chunk had read when the face came
};

struct Chunk {
    const char* begin = nullptr;
    const char* end = nullptr;
    std::vector<float> pos, tex;
    std::vector<long long> cv, ct;  // raw indices as written (1-based or negative), ct = NO_TEX when absent
    std::vector<Face> faces;
    long lines = 0;
    long err_line = 0;              // > 0: a syntax error on that line of the chunk
    const char* err_what = nullptr;
};

void parse_chunk(Chunk* c) {
    const char* p = c->begin;
    const char* const eof = c->end;
    c->pos.reserve(size_t(eof - p) / 24);
    const char* tok_s[64];
    const char* tok_e[64];
    long line_no = 0;
    auto syntax = [&](const char* what) {
        c->err_line = line_no;
        c->err_what = what;
    };
    while (p < eof) {
        const char* le = p;
        while (le < eof && *le != '\n' && *le != '\r') ++le;
        ++line_no;
        const char* q = p;
        p = le < eof ? le + 1 : le;
        if (q == le || *q == '#') continue;
        while (q < le && is_space(*q)) ++q;
        if (q == le) continue;
        const char* ts = q;
        while (q < le && !is_space(*q)) ++q;
        const size_t tl = size_t(q - ts);
        const bool is_v = tl == 1 && ts[0] == 'v', is_vt = tl == 2 && ts[0] == 'v' && ts[1] == 't',
                   is_f = tl == 1 && ts[0] == 'f';
        if (!is_v && !is_vt && !is_f) continue;
        if (is_f) {
            // faces can have any number of corners; first pass: count tokens (a face needs three)
            int ntok = 0;
            const char* r = q;
            while (r < le) {
                while (r < le && is_space(*r)) ++r;
                if (r == le) break;
                while (r < le && !is_space(*r)) ++r;
                ++ntok;
            }
            if (ntok < 3) continue;
            Face f{uint32_t(c->cv.size()), 0u, uint32_t(c->pos.size() / 3), uint32_t(c->tex.size() / 2)};
            r = q;
            while (r < le) {
                while (r < le && is_space(*r)) ++r;
                if (r == le) break;
                const char* s = r;
                while (r < le && !is_space(*r)) ++r;
                const char* e = r;
                const char* s1 = s;
                while (s1 < e && *s1 != '/') ++s1;
                long long vi = 0, ti = NO_TEX;
                if (!parse_index(s, s1, &vi)) return syntax("bad face index");
                if (s1 < e) {
                    const char* s2 = s1 + 1;
                    const char* e2 = s2;
                    while (e2 < e && *e2 != '/') ++e2;
                    if (e2 > s2 && !parse_index(s2, e2, &ti)) return syntax("bad face index");
                }
                c->cv.push_back(vi);
                c->ct.push_back(ti);
            }
            f.count = uint32_t(c->cv.size()) - f.first;
            c->faces.push_back(f);
            continue;
        }
        // "v x y z [w]" / "vt u v [w]"
        const int need = is_v ? 3 : 2;
        int ntok = 0;
        while (q < le && ntok < 64) {
            while (q < le && is_space(*q)) ++q;
            if (q == le) break;
            tok_s[ntok] = q;
            while (q < le && !is_space(*q)) ++q;
            tok_e[ntok++] = q;
        }
        if (ntok < need) continue;
        double v[3];
        for (int i = 0; i < need; ++i)
            if (!parse_double(tok_s[i], tok_e[i], &v[i])) return syntax("could not convert string to float");
        std::vector<float>& dst = is_v ? c->pos : c->tex;
        for (int i = 0; i < need; ++i) dst.push_back(float(v[i]));
    }
    c->lines = line_no;
}

// CPUs this process may really use: the scheduler affinity mask, capped by the cgroup's CPU quota (a container on a
// 256-thread host is typically granted 8 or 16) - std::thread::hardware_concurrency() reports the host
long usable_cpus() {
    long n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = long(std::thread::hardware_concurrency());
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = strtol(quota, nullptr, 10) / period;
            if (q >= 1 && q < n) n = q;
        }
        fclose(f);
    }
    return n < 1 ? 1 : n;
}

constexpr long MAX_READER_THREADS = 16;

// files being parsed right now, process-wide (a folder's ingest runs 1-4 reader threads, each inside mvlm_obj_read): the
// host's cores are shared among them instead of every call starting a full complement (4 x 16 threads on 16 cores)
std::atomic<int> g_active_reads{0};
struct ActiveRead {
    int n;
    ActiveRead() : n(++g_active_reads) {}
    ~ActiveRead() { --g_active_reads; }
};

int reader_threads(size_t bytes, int active_reads) {
    long want = long(bytes / (256u << 10));  // a thread per 256 KB of text, at most MAX_READER_THREADS
    if (const char* e = getenv("MVLM_OBJ_THREADS")) want = strtol(e, nullptr, 10);
    static const long cpus = usable_cpus();
    const long share = cpus / (active_reads < 1 ? 1 : active_reads);  // this call's part of the cores
    if (!getenv("MVLM_OBJ_THREADS") && want > share) want = share;
    return int(want < 1 ? 1 : want > MAX_READER_THREADS ? MAX_READER_THREADS : want);
}

// parse_chunk with the boundary's promise kept: nothing throws out of it (an allocation failure becomes the chunk's error)
void parse_chunk_guarded(Chunk* c) noexcept {
    try {
        parse_chunk(c);
    } catch (const std::exception&) {
        if (c->err_line <= 0) c->err_line = c->lines + 1;
        c->err_what = "out of memory while parsing";
    } catch (...) {
        if (c->err_line <= 0) c->err_line = c->lines + 1;
        c->err_what = "internal error while parsing";
    }
}

}  // namespace

extern "C" int mvlm_obj_read(const char* path, mvlm_obj** out, char* err, int err_len) {
    if (!path || !out) { set_err(err, err_len, "obj_read: bad arguments"); return MVLM_OBJ_ERR_ARGS; }
    *out = nullptr;
    // MVLM_OBJ_TIMING=1: the phases' end times on stderr (diagnostic)
    struct timespec t_begin;
    clock_gettime(CLOCK_MONOTONIC, &t_begin);
    const bool timing = getenv("MVLM_OBJ_TIMING") != nullptr;
    auto lap = [&](const char* what) {
        if (!timing) return;
        struct timespec t;
        clock_gettime(CLOCK_MONOTONIC, &t);
        fprintf(stderr, "  obj_read %-12s %.3f ms\n", what, (t.tv_sec - t_begin.tv_sec) * 1e3 + (t.tv_nsec - t_begin.tv_nsec) * 1e-6);
    };
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, err_len, std::string("File ") + path + " does not exist."); return MVLM_OBJ_ERR_FILE; }
    // (an uninitialised buffer: std::vector would zero-fill 6 MB that fread overwrites at once - 0.3 ms of a 2.5 ms parse)
    struct Text {
        std::unique_ptr<char[]> buf;
        size_t n = 0;
        const char* data() const { return buf.get(); }
        size_t size() const { return n; }
    } text;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz > 0) {
            text.buf.reset(new (std::nothrow) char[size_t(sz)]);
            if (!text.buf) {
                fclose(f);
                set_err(err, err_len, std::string("File ") + path + ": out of memory");
                return MVLM_OBJ_ERR_FILE;
            }
            text.n = fread(text.buf.get(), 1, size_t(sz), f);
        }
        fclose(f);
    }
    lap("read");
    // ---- cut into chunks right behind a line terminator, parse them side by side
    const ActiveRead active;
    const int n_chunks = reader_threads(text.size(), active.n);
    std::vector<Chunk> chunks{size_t(n_chunks)};
    {
        const char* const base = text.data();
        const char* const eof = base + text.size();
        const char* cut = base;
        for (int k = 0; k < n_chunks; ++k) {
            chunks[size_t(k)].begin = cut;
            const char* next = k + 1 == n_chunks ? eof : base + text.size() * size_t(k + 1) / size_t(n_chunks);
            if (next < cut) next = cut;
            while (next < eof && next > base && next[-1] != '\n' && next[-1] != '\r') ++next;
            chunks[size_t(k)].end = cut = next;
        }
        // A worker that cannot be started (EAGAIN under a pids limit: predict_files runs several readers) must not unwind
        // through joinable threads (std::terminate in a process that owns the GPU): its chunk and the remaining ones are
        // parsed on this thread, the workers already running are joined as usual.
        std::vector<std::thread> workers;
        workers.reserve(size_t(n_chunks));
        int started = 1;  // chunks [1, started) have a worker
        for (; started < n_chunks; ++started) {
            try {
                workers.emplace_back(parse_chunk_guarded, &chunks[size_t(started)]);
            } catch (...) {
                break;
            }
        }
        parse_chunk_guarded(&chunks[0]);
        for (int k = started; k < n_chunks; ++k) parse_chunk_guarded(&chunks[size_t(k)]);
        for (auto& w : workers) w.join();
    }
    lap("parsed");
    // ---- the first error in file order is the one a sequential parse stops at
    {
        long lines_before = 0;
        for (const Chunk& c : chunks) {
            if (c.err_line > 0) {
                set_err(err, err_len, std::string("File ") + path + " line " + std::to_string(lines_before + c.err_line) + ": " + c.err_what);
                return MVLM_OBJ_ERR_SYNTAX;
            }
            lines_before += c.lines;
        }
    }
    // ---- element bases, concatenated positions / texture coordinates
    std::vector<long long> pos_base(size_t(n_chunks) + 1, 0), tex_base(size_t(n_chunks) + 1, 0);
    size_t n_corner_refs = 0, n_fan = 0;
    for (int k = 0; k < n_chunks; ++k) {
        pos_base[size_t(k) + 1] = pos_base[size_t(k)] + (long long)(chunks[size_t(k)].pos.size() / 3);
        tex_base[size_t(k) + 1] = tex_base[size_t(k)] + (long long)(chunks[size_t(k)].tex.size() / 2);
        n_corner_refs += chunks[size_t(k)].cv.size();
        for (const Face& fc : chunks[size_t(k)].faces) n_fan += fc.count - 2;
    }
    const int64_t n_pos = pos_base[size_t(n_chunks)], n_tex = tex_base[size_t(n_chunks)];
    if (n_pos == 0) {
        set_err(err, err_len, std::string("File ") + path + " does not contain any points.");
        return MVLM_OBJ_ERR_EMPTY;
    }
    std::vector<float> pos(size_t(n_pos) * 3), tex(size_t(n_tex) * 2);
    for (int k = 0; k < n_chunks; ++k) {
        const Chunk& c = chunks[size_t(k)];
        if (!c.pos.empty()) memcpy(&pos[size_t(pos_base[size_t(k)]) * 3], c.pos.data(), c.pos.size() * sizeof(float));
        if (!c.tex.empty()) memcpy(&tex[size_t(tex_base[size_t(k)]) * 2], c.tex.data(), c.tex.size() * sizeof(float));
    }
    lap("concatenated");
    mvlm_obj* o = new mvlm_obj;
    o->n_positions = n_pos;
    if (n_fan == 0) {  // a point cloud: keep the points, nothing to render or to snap to
        o->verts.swap(pos);
        *out = o;
        return 0;
    }
    // ---- corners (v, vt) numbered in order of first use.  Scans name a point with ONE texture coordinate almost
    // everywhere: the first pair of a point sits in a table indexed by the point, only further pairs go through the hash map
    std::vector<long long> corner_v, corner_t;
    std::vector<long long> first_t(size_t(n_pos), NO_TEX - 1);   // NO_TEX - 1: point not used yet
    std::vector<int32_t> first_id(size_t(n_pos), 0);
    CornerMap corners;
    std::vector<int32_t> tris;
    std::vector<int32_t> ids;
    corner_v.reserve(size_t(n_pos) + 16);
    corner_t.reserve(size_t(n_pos) + 16);
    tris.reserve(n_fan * 3);
    (void)n_corner_refs;
    for (int k = 0; k < n_chunks; ++k) {
        const Chunk& c = chunks[size_t(k)];
        for (const Face& fc : c.faces) {
            const long long seen_pos = pos_base[size_t(k)] + fc.pos_before, seen_tex = tex_base[size_t(k)] + fc.tex_before;
            ids.clear();
            for (uint32_t j = fc.first; j < fc.first + fc.count; ++j) {
                long long vi = c.cv[j], ti = c.ct[j];
                vi = vi > 0 ? vi - 1 : seen_pos + vi;
                ti = ti == NO_TEX ? -1 : (ti > 0 ? ti - 1 : seen_tex + ti);
                // indices outside +-2^31 cannot be valid and are rejected below
                if (vi < -(1ll << 31) || vi >= (1ll << 31)) vi = -(1ll << 31);
                if (ti < -(1ll << 31) || ti >= (1ll << 31)) ti = ti < 0 ? -(1ll << 31) : (1ll << 31) - 1;
                int32_t id;
                if (vi >= 0 && vi < n_pos && first_t[size_t(vi)] == NO_TEX - 1) {
                    first_t[size_t(vi)] = ti;
                    id = first_id[size_t(vi)] = int32_t(corner_v.size());
                    corner_v.push_back(vi);
                    corner_t.push_back(ti);
                } else if (vi >= 0 && vi < n_pos && first_t[size_t(vi)] == ti) {
                    id = first_id[size_t(vi)];
                } else {
                    const uint64_t key = (uint64_t(uint32_t(int32_t(vi))) << 32) | uint64_t(uint32_t(int32_t(ti)));
                    bool fresh = false;
                    id = corners.get(key, int32_t(corner_v.size()), &fresh);
                    if (fresh) { corner_v.push_back(vi); corner_t.push_back(ti); }
                }
                ids.push_back(id);
            }
            for (size_t q = 1; q + 1 < ids.size(); ++q) {  // polygon -> fan
                tris.push_back(ids[0]); tris.push_back(ids[q]); tris.push_back(ids[q + 1]);
            }
        }
    }
    lap("corners");
    const size_t nc = corner_v.size();
    o->verts.resize(nc * 3);
    bool any_uv = false;
    for (size_t i = 0; i < nc; ++i) {
        const long long vi = corner_v[i];
        if (vi < 0 || vi >= n_pos) {
            delete o;
            set_err(err, err_len, std::string("File ") + path + " references a vertex that does not exist.");
            return MVLM_OBJ_ERR_INDEX;
        }
        memcpy(&o->verts[i * 3], &pos[size_t(vi) * 3], 3 * sizeof(float));
        any_uv |= corner_t[i] >= 0;
    }
    if (n_tex > 0 && any_uv) {
        o->uvs.assign(nc * 2, 0.0f);
        for (size_t i = 0; i < nc; ++i) {
            long long ti = corner_t[i];
            if (ti < 0) continue;  // corner without a texture index -> (0, 0)
            if (ti > n_tex - 1) ti = n_tex - 1;
            o->uvs[i * 2] = tex[size_t(ti) * 2];
            o->uvs[i * 2 + 1] = tex[size_t(ti) * 2 + 1];
        }
    }
    o->tris.swap(tris);
    *out = o;
    lap("gathered");
    return 0;
}

extern "C" int mvlm_obj_info(const mvlm_obj* obj, int64_t* n_verts, int64_t* n_tris, int* has_uvs) {
    if (!obj) return MVLM_OBJ_ERR_ARGS;
    if (n_verts) *n_verts = int64_t(obj->verts.size() / 3);
    if (n_tris) *n_tris = int64_t(obj->tris.size() / 3);
    if (has_uvs) *has_uvs = obj->uvs.empty() ? 0 : 1;
    return 0;
}

extern "C" int mvlm_obj_copy(const mvlm_obj* obj, float* verts, float* uvs, int32_t* tris) {
    if (!obj) return MVLM_OBJ_ERR_ARGS;
    if (verts && !obj->verts.empty()) memcpy(verts, obj->verts.data(), obj->verts.size() * sizeof(float));
    if (uvs && !obj->uvs.empty()) memcpy(uvs, obj->uvs.data(), obj->uvs.size() * sizeof(float));
    if (tris && !obj->tris.empty()) memcpy(tris, obj->tris.data(), obj->tris.size() * sizeof(int32_t));
    return 0;
}

extern "C" void mvlm_obj_free(mvlm_obj* obj) { delete obj; }
