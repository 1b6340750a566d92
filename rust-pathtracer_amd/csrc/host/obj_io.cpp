#include "obj_io.h"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <tuple>

namespace pth {
namespace {

std::string dirname_of(const std::string& p) { size_t k = p.find_last_of("/\\"); return k == std::string::npos ? std::string() : p.substr(0, k + 1); }

bool load_mtl(const std::string& path, std::vector<std::string>* names) {
    std::ifstream f(path);
    if (!f) return false;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string tok, name;
        if ((ss >> tok) && tok == "newmtl" && (ss >> name)) names->push_back(name);
    }
    return true;
}

}  // namespace

bool load_obj(const std::string& path, ObjFile* out, std::string* error) {
    std::ifstream f(path);
    if (!f) { *error = "could not find obj file or mtl file " + path; return false; }
    std::vector<float> v, vn;
    std::string group = "unnamed_object";
    int material = -1;
    ObjModel* cur = nullptr;
    std::map<std::tuple<long, std::string, long>, uint32_t> remap;
    std::string line;
    int lineno = 0;
    while (std::getline(f, line)) {
        ++lineno;
        std::istringstream ss(line);
        std::string tok;
        if (!(ss >> tok) || tok[0] == '#') continue;
        if (tok == "v" || tok == "vn") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) { *error = path + ":" + std::to_string(lineno) + ": bad vertex"; return false; }
            std::vector<float>& dst = tok == "v" ? v : vn;
            dst.push_back(x); dst.push_back(y); dst.push_back(z);
        } else if (tok == "o" || tok == "g") {
            if (!(ss >> group)) group = "unnamed_object";
            cur = nullptr;
        } else if (tok == "mtllib") {
            std::string name;
            if (ss >> name) {
                // tobj reports a missing .mtl as an error and the reference expects() it (meshes.rs:30)
                if (!load_mtl(dirname_of(path) + name, &out->materials)) { *error = "Failed to load MTL file " + dirname_of(path) + name; return false; }
            }
        } else if (tok == "usemtl") {
            std::string name;
            ss >> name;
            material = -1;
            for (size_t k = 0; k < out->materials.size(); ++k) if (out->materials[k] == name) material = (int)k;
            cur = nullptr;
        } else if (tok == "f") {
            if (!cur) {
                out->models.emplace_back();
                cur = &out->models.back();
                cur->name = group; cur->material = material;
                remap.clear();
            }
            std::vector<uint32_t> corner;
            std::string c;
            while (ss >> c) {
                std::string parts[3];
                size_t a = c.find('/');
                if (a == std::string::npos) parts[0] = c;
                else {
                    parts[0] = c.substr(0, a);
                    size_t b = c.find('/', a + 1);
                    if (b == std::string::npos) parts[1] = c.substr(a + 1);
                    else { parts[1] = c.substr(a + 1, b - a - 1); parts[2] = c.substr(b + 1); }
                }
                long nv = (long)(v.size() / 3), nn = (long)(vn.size() / 3);
                long vi = std::strtol(parts[0].c_str(), nullptr, 10);
                vi = vi > 0 ? vi - 1 : nv + vi;
                long ni = -1;
                if (!parts[2].empty()) { ni = std::strtol(parts[2].c_str(), nullptr, 10); ni = ni > 0 ? ni - 1 : nn + ni; }
                if (vi < 0 || vi >= nv || (ni >= nn)) { *error = path + ":" + std::to_string(lineno) + ": index out of range"; return false; }
                auto key = std::make_tuple(vi, parts[1], ni);
                auto it = remap.find(key);
                uint32_t idx;
                if (it == remap.end()) {
                    idx = (uint32_t)(cur->positions.size() / 3);
                    remap.emplace(key, idx);
                    for (int k = 0; k < 3; ++k) cur->positions.push_back(v[3 * vi + k]);
                    if (ni >= 0) for (int k = 0; k < 3; ++k) cur->normals.push_back(vn[3 * ni + k]);
                } else idx = it->second;
                corner.push_back(idx);
            }
            for (size_t k = 1; k + 1 < corner.size(); ++k) { cur->indices.push_back(corner[0]); cur->indices.push_back(corner[k]); cur->indices.push_back(corner[k + 1]); }
        }
    }
    return true;
}

}  // namespace pth
