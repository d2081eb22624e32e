// Wavefront OBJ ingest on the host (no device code): what obj_to_actor gets out of vtkOBJReader
// (reference src/mvlm/utils/utils3d.py:16-24) - float32 points, one texture coordinate per point
// (a point is duplicated when it is used with several `vt` indices), polygons as triangle fans,
// `.mtl` / normals / groups ignored.  A 100k-triangle scan is ~6 MB of text; this reader is what
// keeps file ingest (tens of ms) below the GPU time of the views it feeds.
//
// The corner numbering is the contract with mvlm_amd/utils/mesh_io.py:_parse_obj (the pure-Python
// statement of the same rules, kept as the cross-check in tests): corners are numbered in order of
// first use of each (v, vt) pair, negative indices are relative to the elements read so far.
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

}  // namespace

extern "C" int mvlm_obj_read(const char* path, mvlm_obj** out, char* err, int err_len) {
    if (!path || !out) { set_err(err, err_len, "obj_read: bad arguments"); return MVLM_OBJ_ERR_ARGS; }
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) { set_err(err, err_len, std::string("File ") + path + " does not exist."); return MVLM_OBJ_ERR_FILE; }
    std::vector<char> text;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        text.resize(sz > 0 ? size_t(sz) : 0);
        const size_t got = text.empty() ? 0 : fread(text.data(), 1, text.size(), f);
        text.resize(got);
        fclose(f);
    }
    std::vector<float> pos, tex;
    std::vector<long long> corner_v, corner_t;
    std::vector<int32_t> tris;
    std::vector<int32_t> ids;
    CornerMap corners;
    pos.reserve(text.size() / 24);
    const char* p = text.data();
    const char* const eof = p + text.size();
    long line_no = 0;
    const char* tok_s[64];
    const char* tok_e[64];
    auto syntax = [&](const char* what) {
        set_err(err, err_len, std::string("File ") + path + " line " + std::to_string(line_no) + ": " + what);
        return MVLM_OBJ_ERR_SYNTAX;
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
            // faces can have any number of corners
            ids.clear();
            int ntok = 0;
            const char* r = q;
            // first pass: count tokens (a face needs three)
            while (r < le) {
                while (r < le && is_space(*r)) ++r;
                if (r == le) break;
                while (r < le && !is_space(*r)) ++r;
                ++ntok;
            }
            if (ntok < 3) continue;
            const long long n_pos = (long long)(pos.size() / 3), n_tex = (long long)(tex.size() / 2);
            r = q;
            while (r < le) {
                while (r < le && is_space(*r)) ++r;
                if (r == le) break;
                const char* s = r;
                while (r < le && !is_space(*r)) ++r;
                const char* e = r;
                const char* s1 = s;
                while (s1 < e && *s1 != '/') ++s1;
                long long vi = 0, ti = -1;
                if (!parse_index(s, s1, &vi)) return syntax("bad face index");
                vi = vi > 0 ? vi - 1 : n_pos + vi;
                if (s1 < e) {
                    const char* s2 = s1 + 1;
                    const char* e2 = s2;
                    while (e2 < e && *e2 != '/') ++e2;
                    if (e2 > s2) {
                        if (!parse_index(s2, e2, &ti)) return syntax("bad face index");
                        ti = ti > 0 ? ti - 1 : n_tex + ti;
                    }
                }
                // the pair as a 64-bit key; indices outside +-2^31 cannot be valid and are rejected below
                if (vi < -(1ll << 31) || vi >= (1ll << 31)) vi = -(1ll << 31);
                if (ti < -(1ll << 31) || ti >= (1ll << 31)) ti = ti < 0 ? -(1ll << 31) : (1ll << 31) - 1;
                const uint64_t key = (uint64_t(uint32_t(int32_t(vi))) << 32) | uint64_t(uint32_t(int32_t(ti)));
                bool fresh = false;
                const int32_t id = corners.get(key, int32_t(corner_v.size()), &fresh);
                if (fresh) { corner_v.push_back(vi); corner_t.push_back(ti); }
                ids.push_back(id);
            }
            for (size_t k = 1; k + 1 < ids.size(); ++k) {  // polygon -> fan
                tris.push_back(ids[0]); tris.push_back(ids[k]); tris.push_back(ids[k + 1]);
            }
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
        std::vector<float>& dst = is_v ? pos : tex;
        for (int i = 0; i < need; ++i) dst.push_back(float(v[i]));
    }
    const int64_t n_pos = int64_t(pos.size() / 3), n_tex = int64_t(tex.size() / 2);
    if (n_pos == 0) {
        set_err(err, err_len, std::string("File ") + path + " does not contain any points.");
        return MVLM_OBJ_ERR_EMPTY;
    }
    mvlm_obj* o = new mvlm_obj;
    o->n_positions = n_pos;
    if (tris.empty()) {  // a point cloud: keep the points, nothing to render or to snap to
        o->verts.swap(pos);
        *out = o;
        return 0;
    }
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
