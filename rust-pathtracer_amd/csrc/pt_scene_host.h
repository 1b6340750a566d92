// pt_scene_host.h — host-side scene flattening for the HIP engine (see pt_scene_host.cpp).
#ifndef PT_SCENE_HOST_H
#define PT_SCENE_HOST_H
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_blob.h"

namespace pth {

struct HostScene {
    std::vector<uint32_t> blob;   // pt_blob.h layout
    std::vector<float> tex;       // texture texels
    std::vector<pt_camera> cameras;
    std::vector<uint32_t> curve_offsets;
    std::vector<int> mesh_has_light;
    uint32_t light_count = 0, material_count = 0, curve_count = 0;
};

bool build_host_scene(const pt_scene_desc& desc, HostScene* out, std::string* error);

}  // namespace pth
#endif
