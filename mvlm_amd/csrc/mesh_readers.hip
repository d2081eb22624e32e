// Mesh ingest for the formats of the reference's legacy multi-format reader (host only, no device code):
// Utils3D.multi_read_surface, reference src/mvlm/utils/utils3d.py:389-423 - vtkPLYReader, vtkSTLReader,
// vtkPolyDataReader (legacy .vtk) and vtkVRMLImporter (.wrl, the BU-3DFE RAW scans) next to vtkOBJReader.
// Each reader produces what the renderer and the surface snap need and nothing else: float32 points, one
// texture coordinate per point where the file has them, polygons as triangle fans - the same mvlm_obj
// handle the OBJ reader fills (obj_reader.hip), so upload, render and snap do not care about the source.
//
//   .ply  ASCII and binary little/big endian; element vertex with x y z (+ s t | u v | texture_u texture_v),
//         element face with a list property (vertex_indices | vertex_index); other elements / properties skipped
//   .stl  ASCII and binary; vtkSTLReader merges coincident points (Merging is on by default): vertices with
//         bit-identical coordinates share one point, numbered in order of first appearance
//   .vtk  legacy POLYDATA, ASCII and BINARY (big endian): POINTS, POLYGONS, TRIANGLE_STRIPS,
//         POINT_DATA TEXTURE_COORDINATES <name> 2 <type>; the 5.x OFFSETS / CONNECTIVITY form of POLYGONS too
//   .wrl  VRML 2.0 IndexedFaceSet: Coordinate point [], coordIndex [], TextureCoordinate point [],
//         texCoordIndex [] (a point used with several texture coordinates is duplicated, like OBJ corners);
//         the last IndexedFaceSet of the file wins (utils3d.py:401: GetActors().GetLastActor())
// Malformed input returns an error code and a message, never undefined behaviour: every count is checked
// against the bytes that are really there (tests/test_mesh_readers.py runs this file under ASan + UBSan).
#include <cctype>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/mvlm_hip.h"
#include "mesh_obj.h"

namespace {

constexpr int64_t MAX_ELEMS = int64_t(1) << 31;  // nothing in a mesh file may claim more elements than this

void set_err(char* err, int err_len, const std::string& msg) {
    if (err && err_len > 0) snprintf(err, size_t(err_len), "%s", msg.c_str());
}

bool read_file(const char* path, std::vector<char>* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    out->resize(sz > 0 ? size_t(sz) : 0);
    const size_t got = out->empty() ? 0 : fread(out->data(), 1, out->size(), f);
    out->resize(got);
    fclose(f);
    return true;
}

std::string lower_ext(const char* path) {
    const char* dot = strrchr(path, '.');
    const char* slash = strrchr(path, '/');
    if (!dot || (slash && dot < slash)) return "";
    std::string e(dot);
    for (auto& c : e) c = char(tolower((unsigned char)c));
    return e;
}

// whitespace-separated token scanner over a byte range
struct Scan {
    const char* p;
    const char* end;
    bool next(const char** s, const char** e) {
        while (p < end && isspace((unsigned char)*p)) ++p;
        if (p >= end) return false;
        *s = p;
        while (p < end && !isspace((unsigned char)*p)) ++p;
        *e = p;
        return true;
    }
    bool word(std::string* w) {
        const char *s, *e;
        if (!next(&s, &e)) return false;
        w->assign(s, e);
        return true;
    }
    bool number(double* v) {
        const char *s, *e;
        if (!next(&s, &e)) return false;
        char buf[64];
        const size_t n = size_t(e - s);
        if (n == 0 || n >= sizeof buf) return false;
        memcpy(buf, s, n);
        buf[n] = 0;
        char* stop = nullptr;
        errno = 0;
        *v = strtod(buf, &stop);
        return stop == buf + n;
    }
    bool integer(long long* v) {
        double d;
        if (!number(&d)) return false;
        if (!(d > -9e15 && d < 9e15)) return false;
        *v = (long long)d;
        return double(*v) == d;
    }
    void skip_line() {
        while (p < end && *p != '\n') ++p;
        if (p < end) ++p;
    }
};

std::string lower(std::string s) {
    for (auto& c : s) c = char(tolower((unsigned char)c));
    return s;
}

// A parsed number used as an index: strtod accepts "nan", "inf" and 1e300, and casting those to an integer type is
// undefined behaviour (on x86 it yields LLONG_MIN, which then passes a "v >= n" test).  Finite, whole and inside
// [0, 2^31) or -1: every caller's own range check rejects the -1.
long long to_index(double d) {
    if (!std::isfinite(d) || d < 0.0 || d >= 2147483648.0 || d != std::floor(d)) return -1;
    return (long long)d;
}

void fan(const std::vector<long long>& ids, std::vector<int32_t>* tris) {
    for (size_t k = 1; k + 1 < ids.size(); ++k) {
        tris->push_back(int32_t(ids[0]));
        tris->push_back(int32_t(ids[k]));
        tris->push_back(int32_t(ids[k + 1]));
    }
}

// all triangle indices must address existing points
bool indices_ok(const std::vector<int32_t>& tris, int64_t n_points) {
    for (int32_t t : tris)
        if (t < 0 || int64_t(t) >= n_points) return false;
    return true;
}

template <class T>
T load_swapped(const char* p, bool swap) {
    unsigned char b[sizeof(T)];
    memcpy(b, p, sizeof(T));
    if (swap)
        for (size_t i = 0; i < sizeof(T) / 2; ++i) {
            const unsigned char t = b[i];
            b[i] = b[sizeof(T) - 1 - i];
            b[sizeof(T) - 1 - i] = t;
        }
    T v;
    memcpy(&v, b, sizeof(T));
    return v;
}

bool host_is_little_endian() {
    const uint16_t x = 1;
    unsigned char b;
    memcpy(&b, &x, 1);
    return b == 1;
}

// ---- PLY -------------------------------------------------------------------------------------------------
enum PlyType { P_I8, P_U8, P_I16, P_U16, P_I32, P_U32, P_F32, P_F64, P_BAD };

PlyType ply_type(const std::string& t) {
    if (t == "char" || t == "int8") return P_I8;
    if (t == "uchar" || t == "uint8") return P_U8;
    if (t == "short" || t == "int16") return P_I16;
    if (t == "ushort" || t == "uint16") return P_U16;
    if (t == "int" || t == "int32") return P_I32;
    if (t == "uint" || t == "uint32") return P_U32;
    if (t == "float" || t == "float32") return P_F32;
    if (t == "double" || t == "float64") return P_F64;
    return P_BAD;
}
int ply_size(PlyType t) {
    static const int s[] = {1, 1, 2, 2, 4, 4, 4, 8, 0};
    return s[t];
}
double ply_get(const char* p, PlyType t, bool swap) {
    switch (t) {
        case P_I8: return double(load_swapped<int8_t>(p, false));
        case P_U8: return double(load_swapped<uint8_t>(p, false));
        case P_I16: return double(load_swapped<int16_t>(p, swap));
        case P_U16: return double(load_swapped<uint16_t>(p, swap));
        case P_I32: return double(load_swapped<int32_t>(p, swap));
        case P_U32: return double(load_swapped<uint32_t>(p, swap));
        case P_F32: return double(load_swapped<float>(p, swap));
        case P_F64: return load_swapped<double>(p, swap);
        default: return 0.0;
    }
}

struct PlyProp {
    std::string name;
    bool is_list = false;
    PlyType count_type = P_BAD, type = P_BAD;
};
struct PlyElem {
    std::string name;
    long long count = 0;
    std::vector<PlyProp> props;
};

int read_ply(const std::vector<char>& data, mvlm_obj* o, std::string* msg) {
    Scan sc{data.data(), data.data() + data.size()};
    std::string w;
    if (!sc.word(&w) || w != "ply") { *msg = "not a PLY file"; return MVLM_OBJ_ERR_SYNTAX; }
    int format = -1;  // 0 ascii, 1 little, 2 big
    std::vector<PlyElem> elems;
    bool header_done = false;
    while (sc.p < sc.end) {
        // header is line-oriented
        const char* ls = sc.p;
        const char* le = ls;
        while (le < sc.end && *le != '\n') ++le;
        Scan line{ls, le};
        sc.p = le < sc.end ? le + 1 : le;
        if (!line.word(&w)) continue;
        if (w == "format") {
            std::string f;
            if (!line.word(&f)) break;
            format = f == "ascii" ? 0 : f == "binary_little_endian" ? 1 : f == "binary_big_endian" ? 2 : -1;
        } else if (w == "element") {
            PlyElem e;
            if (!line.word(&e.name) || !line.integer(&e.count) || e.count < 0 || e.count > MAX_ELEMS) {
                *msg = "bad PLY element line";
                return MVLM_OBJ_ERR_SYNTAX;
            }
            elems.push_back(e);
        } else if (w == "property") {
            if (elems.empty()) { *msg = "PLY property before any element"; return MVLM_OBJ_ERR_SYNTAX; }
            PlyProp pr;
            std::string t;
            if (!line.word(&t)) break;
            if (t == "list") {
                std::string ct, it;
                if (!line.word(&ct) || !line.word(&it) || !line.word(&pr.name)) break;
                pr.is_list = true;
                pr.count_type = ply_type(ct);
                pr.type = ply_type(it);
                if (pr.count_type == P_BAD || pr.count_type == P_F32 || pr.count_type == P_F64) { *msg = "bad PLY list count type"; return MVLM_OBJ_ERR_SYNTAX; }
            } else {
                pr.type = ply_type(t);
                if (!line.word(&pr.name)) break;
            }
            if (pr.type == P_BAD) { *msg = "unknown PLY property type"; return MVLM_OBJ_ERR_SYNTAX; }
            elems.back().props.push_back(pr);
        } else if (w == "end_header") {
            header_done = true;
            break;
        }
    }
    if (!header_done || format < 0) { *msg = "incomplete PLY header"; return MVLM_OBJ_ERR_SYNTAX; }
    const bool swap = format != 0 && ((format == 1) != host_is_little_endian());
    std::vector<float> uv;
    std::vector<long long> ids;
    for (const PlyElem& e : elems) {
        const bool is_vertex = e.name == "vertex", is_face = e.name == "face";
        int ix = -1, iy = -1, iz = -1, iu = -1, iv = -1, ilist = -1;
        for (int k = 0; k < int(e.props.size()); ++k) {
            const std::string& n = e.props[size_t(k)].name;
            if (is_vertex && !e.props[size_t(k)].is_list) {
                if (n == "x") ix = k;
                else if (n == "y") iy = k;
                else if (n == "z") iz = k;
                else if (n == "s" || n == "u" || n == "texture_u") iu = k;
                else if (n == "t" || n == "v" || n == "texture_v") iv = k;
            }
            if (is_face && e.props[size_t(k)].is_list && (n == "vertex_indices" || n == "vertex_index")) ilist = k;
        }
        if (is_vertex && (ix < 0 || iy < 0 || iz < 0)) { *msg = "PLY vertex element without x y z"; return MVLM_OBJ_ERR_SYNTAX; }
        const bool want_uv = is_vertex && iu >= 0 && iv >= 0;
        if (is_vertex) {
            // a count the file cannot possibly hold must not drive an allocation
            if (e.count > int64_t(data.size())) { *msg = "PLY vertex count exceeds the file size"; return MVLM_OBJ_ERR_SYNTAX; }
            o->verts.reserve(size_t(e.count) * 3);
            if (want_uv) uv.reserve(size_t(e.count) * 2);
        }
        for (long long r = 0; r < e.count; ++r) {
            double vals[5] = {0, 0, 0, 0, 0};
            for (int k = 0; k < int(e.props.size()); ++k) {
                const PlyProp& pr = e.props[size_t(k)];
                long long n = 1;
                if (pr.is_list) {
                    if (format == 0) {
                        if (!sc.integer(&n)) { *msg = "truncated PLY body"; return MVLM_OBJ_ERR_SYNTAX; }
                    } else {
                        const int cs = ply_size(pr.count_type);
                        if (sc.end - sc.p < cs) { *msg = "truncated PLY body"; return MVLM_OBJ_ERR_SYNTAX; }
                        n = (long long)ply_get(sc.p, pr.count_type, swap);
                        sc.p += cs;
                    }
                    if (n < 0 || n > (1 << 20)) { *msg = "bad PLY list length"; return MVLM_OBJ_ERR_SYNTAX; }
                    if (k == ilist) ids.clear();
                }
                for (long long j = 0; j < n; ++j) {
                    double v;
                    if (format == 0) {
                        if (!sc.number(&v)) { *msg = "truncated PLY body"; return MVLM_OBJ_ERR_SYNTAX; }
                    } else {
                        const int ts = ply_size(pr.type);
                        if (sc.end - sc.p < ts) { *msg = "truncated PLY body"; return MVLM_OBJ_ERR_SYNTAX; }
                        v = ply_get(sc.p, pr.type, swap);
                        sc.p += ts;
                    }
                    if (pr.is_list) {
                        if (k == ilist) ids.push_back((v >= -1e15 && v <= 1e15) ? (long long)v : -1);
                    } else if (is_vertex) {
                        if (k == ix) vals[0] = v;
                        else if (k == iy) vals[1] = v;
                        else if (k == iz) vals[2] = v;
                        else if (k == iu) vals[3] = v;
                        else if (k == iv) vals[4] = v;
                    }
                }
            }
            if (is_vertex) {
                o->verts.push_back(float(vals[0]));
                o->verts.push_back(float(vals[1]));
                o->verts.push_back(float(vals[2]));
                if (want_uv) {
                    uv.push_back(float(vals[3]));
                    uv.push_back(float(vals[4]));
                }
            } else if (is_face && ilist >= 0 && ids.size() >= 3) {
                for (long long id : ids)
                    if (id < 0 || id >= MAX_ELEMS) { *msg = "PLY face references a vertex that does not exist"; return MVLM_OBJ_ERR_INDEX; }
                fan(ids, &o->tris);
            }
        }
    }
    o->uvs.swap(uv);
    return 0;
}

// ---- STL -------------------------------------------------------------------------------------------------
struct Key3 {
    uint32_t a, b, c;
    bool operator==(const Key3& o) const { return a == o.a && b == o.b && c == o.c; }
};
struct Key3Hash {
    size_t operator()(const Key3& k) const {
        uint64_t h = (uint64_t(k.a) * 0x9E3779B97F4A7C15ull) ^ (uint64_t(k.b) << 21) ^ (uint64_t(k.c) * 0xC2B2AE3D27D4EB4Full);
        h ^= h >> 29;
        return size_t(h * 0xBF58476D1CE4E5B9ull);
    }
};

struct PointMerger {  // vtkSTLReader's point merging: coincident points become one
    std::unordered_map<Key3, int32_t, Key3Hash> map;
    std::vector<float>* verts;
    int32_t add(float x, float y, float z) {
        // +0 and -0 are the same location
        if (x == 0.f) x = 0.f;
        if (y == 0.f) y = 0.f;
        if (z == 0.f) z = 0.f;
        Key3 k;
        memcpy(&k.a, &x, 4);
        memcpy(&k.b, &y, 4);
        memcpy(&k.c, &z, 4);
        auto it = map.find(k);
        if (it != map.end()) return it->second;
        const int32_t id = int32_t(verts->size() / 3);
        verts->push_back(x);
        verts->push_back(y);
        verts->push_back(z);
        map.emplace(k, id);
        return id;
    }
};

int read_stl(const std::vector<char>& data, mvlm_obj* o, std::string* msg) {
    PointMerger pm;
    pm.verts = &o->verts;
    // binary: 80-byte header, uint32 count, count x 50 bytes - decided by the size, as VTK does by content
    if (data.size() >= 84) {
        const uint32_t n = load_swapped<uint32_t>(data.data() + 80, !host_is_little_endian());
        if (uint64_t(n) * 50 + 84 == uint64_t(data.size()) && n > 0) {
            const bool swap = !host_is_little_endian();
            for (uint32_t t = 0; t < n; ++t) {
                const char* p = data.data() + 84 + size_t(t) * 50 + 12;
                int32_t id[3];
                for (int k = 0; k < 3; ++k)
                    id[k] = pm.add(load_swapped<float>(p + 12 * k, swap), load_swapped<float>(p + 12 * k + 4, swap),
                                   load_swapped<float>(p + 12 * k + 8, swap));
                o->tris.push_back(id[0]);
                o->tris.push_back(id[1]);
                o->tris.push_back(id[2]);
            }
            return 0;
        }
    }
    Scan sc{data.data(), data.data() + data.size()};
    std::string w;
    if (!sc.word(&w) || lower(w) != "solid") { *msg = "not an STL file"; return MVLM_OBJ_ERR_SYNTAX; }
    std::vector<int32_t> loop;
    while (sc.word(&w)) {
        w = lower(w);
        if (w == "vertex") {
            double x, y, z;
            if (!sc.number(&x) || !sc.number(&y) || !sc.number(&z)) { *msg = "bad STL vertex"; return MVLM_OBJ_ERR_SYNTAX; }
            loop.push_back(pm.add(float(x), float(y), float(z)));
        } else if (w == "endloop") {
            for (size_t k = 1; k + 1 < loop.size(); ++k) {
                o->tris.push_back(loop[0]);
                o->tris.push_back(loop[k]);
                o->tris.push_back(loop[k + 1]);
            }
            loop.clear();
        }
    }
    return 0;
}

// ---- legacy VTK POLYDATA ---------------------------------------------------------------------------------
int vtk_type_size(const std::string& t) {
    const std::string s = lower(t);
    if (s == "float") return 4;
    if (s == "double") return 8;
    if (s == "int" || s == "unsigned_int") return 4;
    if (s == "long" || s == "unsigned_long" || s == "vtktypeint64" || s == "vtkidtype") return 8;
    if (s == "short" || s == "unsigned_short") return 2;
    if (s == "char" || s == "unsigned_char") return 1;
    return 0;
}

// n values of `type` -> doubles, ASCII or big-endian BINARY
bool vtk_values(Scan* sc, bool binary, const std::string& type, long long n, std::vector<double>* out) {
    const std::string t = lower(type);
    const int sz = vtk_type_size(t);
    if (sz == 0 || n < 0 || n > MAX_ELEMS) return false;
    out->clear();
    if (!binary) {
        if (n > sc->end - sc->p) return false;  // at least one byte per value
        out->reserve(size_t(n));
        for (long long i = 0; i < n; ++i) {
            double v;
            if (!sc->number(&v)) return false;
            out->push_back(v);
        }
        return true;
    }
    if (sc->p < sc->end && *sc->p == '\n') ++sc->p;  // the newline that ends the keyword line
    if ((sc->end - sc->p) / sz < n) return false;
    out->reserve(size_t(n));
    const bool swap = host_is_little_endian();  // legacy binary is big endian
    for (long long i = 0; i < n; ++i, sc->p += sz) {
        double v = 0;
        if (t == "float") v = load_swapped<float>(sc->p, swap);
        else if (t == "double") v = load_swapped<double>(sc->p, swap);
        else if (sz == 4) v = t[0] == 'u' ? double(load_swapped<uint32_t>(sc->p, swap)) : double(load_swapped<int32_t>(sc->p, swap));
        else if (sz == 8) v = double(load_swapped<int64_t>(sc->p, swap));
        else if (sz == 2) v = t[0] == 'u' ? double(load_swapped<uint16_t>(sc->p, swap)) : double(load_swapped<int16_t>(sc->p, swap));
        else v = t[0] == 'u' ? double(load_swapped<uint8_t>(sc->p, false)) : double(load_swapped<int8_t>(sc->p, false));
        out->push_back(v);
    }
    return true;
}

int read_vtk(const std::vector<char>& data, mvlm_obj* o, std::string* msg) {
    Scan sc{data.data(), data.data() + data.size()};
    if (data.size() < 14 || memcmp(data.data(), "# vtk DataFile", 14) != 0) { *msg = "not a legacy VTK file"; return MVLM_OBJ_ERR_SYNTAX; }
    sc.skip_line();  // version
    sc.skip_line();  // title
    std::string w;
    if (!sc.word(&w)) { *msg = "truncated VTK header"; return MVLM_OBJ_ERR_SYNTAX; }
    const std::string mode = lower(w);
    if (mode != "ascii" && mode != "binary") { *msg = "VTK file is neither ASCII nor BINARY"; return MVLM_OBJ_ERR_SYNTAX; }
    const bool binary = mode == "binary";
    std::vector<double> vals, offs;
    std::vector<long long> ids;
    long long n_points = -1;
    bool point_data = false;
    while (sc.word(&w)) {
        const std::string kw = lower(w);
        if (kw == "dataset") {
            if (!sc.word(&w) || lower(w) != "polydata") { *msg = "VTK dataset is not POLYDATA"; return MVLM_OBJ_ERR_SYNTAX; }
        } else if (kw == "points") {
            std::string type;
            if (!sc.integer(&n_points) || !sc.word(&type) || n_points < 0 || n_points > MAX_ELEMS / 3 ||
                !vtk_values(&sc, binary, type, n_points * 3, &vals)) {
                *msg = "bad VTK POINTS section";
                return MVLM_OBJ_ERR_SYNTAX;
            }
            o->verts.resize(vals.size());
            for (size_t i = 0; i < vals.size(); ++i) o->verts[i] = float(vals[i]);
        } else if (kw == "polygons" || kw == "triangle_strips") {
            long long n, size;
            if (!sc.integer(&n) || !sc.integer(&size) || n < 0 || size < 0) { *msg = "bad VTK cell section"; return MVLM_OBJ_ERR_SYNTAX; }
            const char* save = sc.p;
            std::string maybe;
            const bool v5 = sc.word(&maybe) && lower(maybe) == "offsets";
            if (v5) {  // 5.x: "POLYGONS <n_offsets> <n_conn>", OFFSETS type, values, CONNECTIVITY type, values
                std::string type;
                if (!sc.word(&type) || !vtk_values(&sc, binary, type, n, &offs)) { *msg = "bad VTK OFFSETS"; return MVLM_OBJ_ERR_SYNTAX; }
                if (!sc.word(&maybe) || lower(maybe) != "connectivity" || !sc.word(&type) ||
                    !vtk_values(&sc, binary, type, size, &vals)) {
                    *msg = "bad VTK CONNECTIVITY";
                    return MVLM_OBJ_ERR_SYNTAX;
                }
                for (size_t c = 0; c + 1 < offs.size(); ++c) {
                    const long long a = to_index(offs[c]), b = to_index(offs[c + 1]);
                    if (a < 0 || b < a || b > (long long)vals.size()) { *msg = "bad VTK cell offsets"; return MVLM_OBJ_ERR_SYNTAX; }
                    ids.clear();
                    for (long long k = a; k < b; ++k) {
                        const long long id = to_index(vals[size_t(k)]);
                        if (id < 0) { *msg = "VTK cell references a point that does not exist"; return MVLM_OBJ_ERR_INDEX; }
                        ids.push_back(id);
                    }
                    if (kw == "polygons") fan(ids, &o->tris);
                    else
                        for (size_t k = 0; k + 2 < ids.size(); ++k) {  // strip: alternate the winding
                            o->tris.push_back(int32_t(ids[k]));
                            o->tris.push_back(int32_t(ids[k + 1 + (k & 1)]));
                            o->tris.push_back(int32_t(ids[k + 2 - (k & 1)]));
                        }
                }
            } else {
                sc.p = save;
                if (!vtk_values(&sc, binary, "int", size, &vals)) { *msg = "bad VTK cell list"; return MVLM_OBJ_ERR_SYNTAX; }
                size_t i = 0;
                for (long long c = 0; c < n && i < vals.size(); ++c) {
                    const long long m = to_index(vals[i++]);
                    if (m < 0 || i + size_t(m) > vals.size()) { *msg = "bad VTK cell list"; return MVLM_OBJ_ERR_SYNTAX; }
                    ids.clear();
                    for (long long k = 0; k < m; ++k) {
                        const long long id = to_index(vals[i++]);
                        if (id < 0) { *msg = "VTK cell references a point that does not exist"; return MVLM_OBJ_ERR_INDEX; }
                        ids.push_back(id);
                    }
                    if (kw == "polygons") fan(ids, &o->tris);
                    else
                        for (size_t k = 0; k + 2 < ids.size(); ++k) {
                            o->tris.push_back(int32_t(ids[k]));
                            o->tris.push_back(int32_t(ids[k + 1 + (k & 1)]));
                            o->tris.push_back(int32_t(ids[k + 2 - (k & 1)]));
                        }
                }
            }
        } else if (kw == "point_data") {
            sc.skip_line();
            point_data = true;
        } else if (kw == "cell_data") {
            sc.skip_line();
            point_data = false;
        } else if (kw == "texture_coordinates" && point_data) {
            std::string name, type;
            long long dim;
            if (!sc.word(&name) || !sc.integer(&dim) || !sc.word(&type) || dim < 1 || dim > 3 || n_points < 0 ||
                !vtk_values(&sc, binary, type, n_points * dim, &vals)) {
                *msg = "bad VTK TEXTURE_COORDINATES";
                return MVLM_OBJ_ERR_SYNTAX;
            }
            if (dim >= 2) {
                o->uvs.resize(size_t(n_points) * 2);
                for (long long i = 0; i < n_points; ++i) {
                    o->uvs[size_t(i) * 2] = float(vals[size_t(i * dim)]);
                    o->uvs[size_t(i) * 2 + 1] = float(vals[size_t(i * dim + 1)]);
                }
            }
        }
        // every other keyword (VERTICES, LINES, SCALARS, NORMALS, ...): its values are skipped as tokens, which is
        // safe for ASCII; an unknown BINARY block would desynchronise the scan, so stop at the first one
        else if (binary && (kw == "vertices" || kw == "lines" || kw == "scalars" || kw == "normals" || kw == "vectors" ||
                            kw == "color_scalars" || kw == "field" || kw == "lookup_table" || kw == "tensors")) {
            break;
        }
    }
    if (n_points < 0) { *msg = "VTK file without POINTS"; return MVLM_OBJ_ERR_SYNTAX; }
    return 0;
}

// ---- VRML 2.0 IndexedFaceSet -----------------------------------------------------------------------------
// tokens with [ ] { } , as separators; '#' starts a comment
struct VrmlScan {
    const char* p;
    const char* end;
    bool next(std::string* t) {
        for (;;) {
            while (p < end && (isspace((unsigned char)*p) || *p == ',')) ++p;
            if (p < end && *p == '#') {
                while (p < end && *p != '\n') ++p;
                continue;
            }
            break;
        }
        if (p >= end) return false;
        if (*p == '[' || *p == ']' || *p == '{' || *p == '}') {
            t->assign(1, *p++);
            return true;
        }
        if (*p == '"') {  // string literal (url "...")
            const char* s = ++p;
            while (p < end && *p != '"') ++p;
            t->assign(s, p);
            if (p < end) ++p;
            return true;
        }
        const char* s = p;
        while (p < end && !isspace((unsigned char)*p) && *p != ',' && *p != '[' && *p != ']' && *p != '{' && *p != '}' && *p != '#') ++p;
        t->assign(s, p);
        return true;
    }
    // "[ n n n ... ]" (or a single value) -> doubles
    bool numbers(std::vector<double>* out) {
        out->clear();
        std::string t;
        if (!next(&t)) return false;
        auto conv = [&](const std::string& s, double* v) {
            if (s.empty() || s.size() > 63) return false;
            char* stop = nullptr;
            *v = strtod(s.c_str(), &stop);
            return stop == s.c_str() + s.size();
        };
        if (t != "[") {
            double v;
            if (!conv(t, &v)) return false;
            out->push_back(v);
            return true;
        }
        while (next(&t)) {
            if (t == "]") return true;
            double v;
            if (!conv(t, &v)) return false;
            out->push_back(v);
        }
        return false;
    }
};

int read_wrl(const std::vector<char>& data, mvlm_obj* o, std::string* msg) {
    if (data.size() < 5 || memcmp(data.data(), "#VRML", 5) != 0) { *msg = "not a VRML file"; return MVLM_OBJ_ERR_SYNTAX; }
    VrmlScan sc{data.data(), data.data() + data.size()};
    sc.p += 5;
    while (sc.p < sc.end && *sc.p != '\n') ++sc.p;  // rest of the "#VRML V2.0 utf8" line
    struct Set {
        std::vector<double> pts, tex, idx, tidx;
    };
    Set cur, last;
    bool have_last = false;
    std::vector<std::string> stack;  // node types of the open braces
    int set_depth = -1;
    std::string t, prev;
    std::vector<double> tmp;
    auto in_set = [&]() { return set_depth >= 0; };
    while (sc.next(&t)) {
        if (t == "{") {
            if (stack.size() > 256) { *msg = "VRML nesting too deep"; return MVLM_OBJ_ERR_SYNTAX; }
            stack.push_back(prev);
            if (prev == "IndexedFaceSet" && !in_set()) {
                set_depth = int(stack.size());
                cur = Set();
            }
        } else if (t == "}") {
            if (stack.empty()) { *msg = "unbalanced braces in VRML file"; return MVLM_OBJ_ERR_SYNTAX; }
            if (int(stack.size()) == set_depth) {
                if (!cur.pts.empty()) {
                    last = cur;
                    have_last = true;
                }
                set_depth = -1;
            }
            stack.pop_back();
        } else if (in_set() && (t == "point" || t == "coordIndex" || t == "texCoordIndex")) {
            if (!sc.numbers(&tmp)) { *msg = "bad VRML number list"; return MVLM_OBJ_ERR_SYNTAX; }
            if (t == "coordIndex") cur.idx = tmp;
            else if (t == "texCoordIndex") cur.tidx = tmp;
            else if (!stack.empty() && stack.back() == "Coordinate") cur.pts = tmp;
            else if (!stack.empty() && stack.back() == "TextureCoordinate") cur.tex = tmp;
        }
        prev = t;
    }
    if (!have_last) { *msg = "VRML file without an IndexedFaceSet"; return MVLM_OBJ_ERR_EMPTY; }
    const long long n_pts = (long long)(last.pts.size() / 3), n_tex = (long long)(last.tex.size() / 2);
    const bool per_corner_tex = n_tex > 0 && !last.tidx.empty() && last.tidx.size() == last.idx.size();
    const bool per_point_tex = n_tex > 0 && !per_corner_tex && n_tex == n_pts;
    // corners (point, texture coordinate) in order of first use, as the OBJ reader numbers them
    std::unordered_map<uint64_t, int32_t> corner;
    std::vector<long long> ids;
    auto flush = [&]() {
        if (ids.size() >= 3) fan(ids, &o->tris);
        ids.clear();
    };
    for (size_t i = 0; i < last.idx.size(); ++i) {
        const double dv = last.idx[i];
        if (dv < 0) {  // -1 closes a face (NaN is not < 0: it goes on to the index check)
            flush();
            continue;
        }
        const long long v = to_index(dv);
        if (v < 0 || v >= n_pts) { *msg = "VRML face references a point that does not exist"; return MVLM_OBJ_ERR_INDEX; }
        long long tc = -1;
        if (per_corner_tex) {
            tc = to_index(last.tidx[i]);
            if (tc < 0 || tc >= n_tex) { *msg = "VRML face references a texture coordinate that does not exist"; return MVLM_OBJ_ERR_INDEX; }
        } else if (per_point_tex) {
            tc = v;
        }
        const uint64_t key = (uint64_t(uint32_t(v)) << 32) | uint64_t(uint32_t(int32_t(tc)));
        auto it = corner.find(key);
        int32_t id;
        if (it == corner.end()) {
            id = int32_t(o->verts.size() / 3);
            corner.emplace(key, id);
            for (int k = 0; k < 3; ++k) o->verts.push_back(float(last.pts[size_t(v) * 3 + size_t(k)]));
            if (per_corner_tex || per_point_tex) {
                o->uvs.push_back(float(last.tex[size_t(tc) * 2]));
                o->uvs.push_back(float(last.tex[size_t(tc) * 2 + 1]));
            }
        } else {
            id = it->second;
        }
        ids.push_back(id);
    }
    flush();
    if (o->tris.empty()) {  // a point cloud
        o->verts.resize(size_t(n_pts) * 3);
        for (size_t i = 0; i < o->verts.size(); ++i) o->verts[i] = float(last.pts[i]);
        o->uvs.clear();
    }
    return 0;
}

}  // namespace

// Any of the reference's surface formats by file extension (utils3d.py:389-423): .obj .ply .stl .vtk .wrl.
// Same handle, accessors and error conventions as mvlm_obj_read; an unknown extension is MVLM_OBJ_ERR_ARGS
// ("Can not read files with extenstion ...", utils3d.py:421-422).
extern "C" int mvlm_mesh_read(const char* path, mvlm_obj** out, char* err, int err_len) {
    if (!path || !out) { set_err(err, err_len, "mesh_read: bad arguments"); return MVLM_OBJ_ERR_ARGS; }
    *out = nullptr;
    const std::string ext = lower_ext(path);
    if (ext == ".obj") return mvlm_obj_read(path, out, err, err_len);
    if (ext != ".ply" && ext != ".stl" && ext != ".vtk" && ext != ".wrl") {
        set_err(err, err_len, std::string("Can not read files with extension ") + ext);
        return MVLM_OBJ_ERR_ARGS;
    }
    std::vector<char> data;
    if (!read_file(path, &data)) { set_err(err, err_len, std::string("File ") + path + " does not exist."); return MVLM_OBJ_ERR_FILE; }
    mvlm_obj* o = new mvlm_obj;
    std::string msg;
    int rc = ext == ".ply" ? read_ply(data, o, &msg) : ext == ".stl" ? read_stl(data, o, &msg) : ext == ".vtk" ? read_vtk(data, o, &msg)
                                                                                                               : read_wrl(data, o, &msg);
    const int64_t n_points = int64_t(o->verts.size() / 3);
    if (!rc && n_points == 0) {
        msg = "does not contain any points.";
        rc = MVLM_OBJ_ERR_EMPTY;
    }
    if (!rc && !indices_ok(o->tris, n_points)) {
        msg = "references a vertex that does not exist.";
        rc = MVLM_OBJ_ERR_INDEX;
    }
    if (!rc && !o->uvs.empty() && o->uvs.size() != size_t(n_points) * 2) o->uvs.clear();  // partial texture coordinates: none
    if (rc) {
        delete o;
        set_err(err, err_len, std::string("File ") + path + ": " + msg);
        return rc;
    }
    o->n_positions = n_points;
    *out = o;
    return 0;
}
