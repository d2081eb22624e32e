// The host-side mesh handle behind mvlm_obj_read / mvlm_mesh_read (include/mvlm_hip.h), shared by the readers.
#ifndef MVLM_MESH_OBJ_H
#define MVLM_MESH_OBJ_H
#include <cstdint>
#include <vector>

struct mvlm_obj {
    std::vector<float> verts;   // [V,3] corner-expanded (or the raw points for a point cloud)
    std::vector<float> uvs;     // [V,2] or empty
    std::vector<int32_t> tris;  // [T,3]
    int64_t n_positions = 0;
};
#endif
