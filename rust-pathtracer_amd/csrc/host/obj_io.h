// obj_io.h — Wavefront OBJ/MTL reader with the semantics the reference gets from tobj with
// LoadOptions{single_index: true, triangulate: true} (src/parsing/meshes.rs:17-53).
#ifndef PT_OBJ_IO_H
#define PT_OBJ_IO_H
#include <cstdint>
#include <string>
#include <vector>

namespace pth {

struct ObjModel {
    std::string name;
    int material = -1;                 // index into ObjFile::materials (tobj Mesh::material_id), -1 = none
    std::vector<float> positions;      // xyz per unique (v, vt, vn) triple, in order of first use
    std::vector<float> normals;        // xyz per vertex, empty when the faces carry no normals
    std::vector<uint32_t> indices;     // 3 per triangle; polygons are fan-triangulated
};
struct ObjFile {
    std::vector<ObjModel> models;           // one per o/g group and per usemtl change inside it
    std::vector<std::string> materials;     // newmtl names in .mtl order
};

bool load_obj(const std::string& path, ObjFile* out, std::string* error);

}  // namespace pth
#endif
