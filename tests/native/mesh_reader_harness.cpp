// Sanitizer harness for the host-only multi-format mesh readers (built by tests/test_mesh_readers.py with
// g++ -fsanitize=address,undefined; GPU sanitizers are not available, the CPU build is what gets checked).
#include <cstdio>
#include <vector>

#include "../../include/mvlm_hip.h"

int main(int argc, char** argv) {
    for (int i = 1; i < argc; ++i) {
        mvlm_obj* obj = nullptr;
        char err[256] = {0};
        const int rc = mvlm_mesh_read(argv[i], &obj, err, sizeof err);
        if (rc != 0) {
            printf("%s: rc=%d %s\n", argv[i], rc, err);
            continue;
        }
        int64_t nv = 0, nt = 0;
        int has_uv = 0;
        mvlm_obj_info(obj, &nv, &nt, &has_uv);
        std::vector<float> v(size_t(nv) * 3 + 1), uv(size_t(nv) * 2 + 1);
        std::vector<int32_t> t(size_t(nt) * 3 + 1);
        mvlm_obj_copy(obj, v.data(), has_uv ? uv.data() : nullptr, t.data());
        long bad = 0;
        for (int64_t k = 0; k < nt * 3; ++k) bad += (t[size_t(k)] < 0 || t[size_t(k)] >= nv);
        printf("%s: ok verts=%lld tris=%lld uv=%d bad_indices=%ld\n", argv[i], (long long)nv, (long long)nt, has_uv, bad);
        mvlm_obj_free(obj);
    }
    return 0;
}
