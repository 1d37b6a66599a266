// mcrt_flatten.h - host side of mcrt_upload_scene: scene validation, light
// cache, per-primitive octree build and packing into the two word images of
// mcrt_layout.h.
#pragma once
#include <string>
#include <vector>

#include "mcrt.h"
#include "mcrt_layout.h"

namespace mcrt {

struct FlatScene {
    std::vector<uint32_t> small_img;
    std::vector<uint32_t> big_img;
    SceneView view{};            // pointers left null; filled after upload
    uint32_t tree_depth = 0;
    uint32_t ref_count = 0;
    // kept on the host for mcrt_octree_download
    std::vector<float> node_box;       // node_count * 6
    std::vector<uint32_t> node_words;  // node_count * 2
    std::vector<uint32_t> refs;
    float root_cell[6] = { 0, 0, 0, 0, 0, 0 };   // min xyz, max xyz of the root cell
    float pad = 0.0f;
    // build_tree == false (the octree is built on the device, csrc/mcrt_build.h): big_img holds only the triangle and pair
    // records, laid out from word 0 (view.off_tri == 0), and these are the builder's parameters as the host would use them
    bool tree_built = true;
    uint32_t max_leaf = 4, max_depth = 10;
    float stay = 1.0f, pad_rel = 1.0e-4f;
};

// Returns MCRT_OK or an MCRT_ERR_* code with `err` set.
int flatten_scene(const mcrt_scene_desc* scene, const mcrt_accel_params* accel, FlatScene& out, std::string& err, bool build_tree = true);

// Oren-Nayar A and B of Material::BRDF (source/core/material.h:29-33) with the reference's float operations.
void oren_nayar_terms(float roughness, float& A, float& B);

// Camera::Camera + Camera::SetView (source/core/camera.h:19-34) with glm's operation order.
void camera_look_at(uint32_t width, uint32_t height, float fov_y_degrees, const float position[3], const float look_at[3],
                    const float up[3], mcrt_camera* out);

} // namespace mcrt
