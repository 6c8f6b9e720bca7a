// rt_host.hpp - C++ host side above the C ABI, shaped after the reference's own call sites, for
// callers written in a compiled language (the reference is Rust; its toolchain is not in the build
// image, so the Rust shim under ffi/ is source only and this header is the compiled, tested mirror).
//
//   rt_host::Camera      src/camera.rs:43-122   position/yaw/pitch, per-scene start poses, to_uniform
//   rt_host::RayTracer   src/renderer.rs:40-351 (new: scene in, grid built) and :800-868 (render);
//                        src/core/ray_tracing_layer.rs:155,436 (ComputeState::new / ::render -> Vec<u8>)
//
// Error behaviour follows the reference: construction and render return errors to the caller
// (Result<_, Box<dyn Error>> there, std::runtime_error here); nothing falls back to the CPU.
#pragma once

#include <cmath>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_mi355.h"

namespace rt_host {

struct Camera {
    float position[3];
    float yaw, pitch;

    // src/camera.rs:59-66
    static Camera for_scene(const std::string& scene) {
        const float pi = 3.14159265358979323846f;
        if (scene == "composed") return {{0.0f, 40.0f, 40.0f}, pi, -0.7f};
        if (scene == "walls") return {{0.0f, 5.0f, 0.0f}, 0.0f, 0.0f};
        if (scene == "tunnel") return {{0.0f, 0.0f, 20.0f}, pi, 0.0f};
        if (scene == "gltf") return {{200.0f, 200.0f, 300.0f}, 3.35f, -0.28f};
        if (scene == "pyramid") return {{0.0f, 8.0f, 20.0f}, pi, -0.5f};
        return {{0.0f, 8.0f, 15.0f}, pi, -0.6f};
    }

    // src/camera.rs:104-122
    CameraUniform to_uniform(float time, float screen_height, float fov = 0.785398f, bool show_grid = false) const {
        CameraUniform u;
        rt_camera_to_uniform(position, yaw, pitch, time, screen_height, fov, show_grid ? 1 : 0, &u);
        return u;
    }
};

struct Frame {
    uint32_t width = 0, height = 0;
    std::vector<rt_hit> primary;  // row-major, id == RT_MISS where the ray hit nothing
    std::vector<rt_hit> shadow;   // empty when shadows were not requested
};

class RayTracer {
public:
    // RayTracer::new (src/renderer.rs:55-99): takes the scene, builds the grid (here: on the GPU).
    RayTracer(const std::vector<BoxData>& boxes, const std::vector<TriangleData>& triangles, uint32_t width,
              uint32_t height, int device_id = 0, const rt_params* params = nullptr)
        : width_(width), height_(height) {
        if (rt_create(&ctx_, device_id, params) != RT_OK) throw std::runtime_error(rt_create_error());
        if (params)
            params_ = *params;
        else
            rt_default_params(&params_);
        try {
            check(rt_set_scene(ctx_, boxes.data(), (uint32_t)boxes.size(), triangles.data(), (uint32_t)triangles.size()));
            check(rt_build_grid(ctx_, 0.0f, 0, RT_GRID_HIERARCHICAL));
        } catch (...) {
            rt_destroy(ctx_);
            throw;
        }
    }
    ~RayTracer() { rt_destroy(ctx_); }
    RayTracer(const RayTracer&) = delete;
    RayTracer& operator=(const RayTracer&) = delete;

    // RayTracer::render (src/renderer.rs:800-868) / ComputeState::render (ray_tracing_layer.rs:436-493):
    // camera uniform from the pose, one dispatch, blocking read-back.
    Frame render(const Camera& camera, float time, bool shadows = true) const {
        const CameraUniform u = camera.to_uniform(time, (float)height_, params_.fov, false);
        Frame f;
        f.width = width_;
        f.height = height_;
        f.primary.resize((size_t)width_ * height_);
        if (shadows) f.shadow.resize((size_t)width_ * height_);
        check(rt_trace_primary(ctx_, &u, width_, height_, nullptr, f.primary.data(), shadows ? f.shadow.data() : nullptr));
        return f;
    }

    // The bytes ComputeState::render returns (src/core/ray_tracing_layer.rs:436-493): the shader's Rgba8Unorm
    // picture, width * height * 4, row-major.
    std::vector<uint8_t> render_rgba(const Camera& camera, float time, const rt_render_opts* opts = nullptr) const {
        const CameraUniform u = camera.to_uniform(time, (float)height_, params_.fov, false);
        std::vector<uint8_t> rgba((size_t)width_ * height_ * 4);
        check(rt_render(ctx_, &u, width_, height_, nullptr, opts, rgba.data()));
        return rgba;
    }

    // The "Materials" storage buffer (src/renderer.rs:147-156).
    void set_materials(const std::vector<MaterialData>& materials) {
        check(rt_set_materials(ctx_, materials.data(), (uint32_t)materials.size()));
    }

    // The texture array (src/renderer.rs:158-259) from the loader's TextureData records.
    void set_textures(const std::vector<rt_texture>& textures) {
        check(rt_set_textures(ctx_, textures.empty() ? nullptr : textures.data(), (uint32_t)textures.size()));
    }

    GridMetadata grid_metadata() const {
        GridMetadata m;
        check(rt_get_grid(ctx_, &m, nullptr, nullptr));
        return m;
    }
    rt_ctx* handle() const { return ctx_; }

private:
    void check(int rc) const {
        if (rc != RT_OK) throw std::runtime_error(rt_last_error(ctx_));
    }
    rt_ctx* ctx_ = nullptr;
    rt_params params_;
    uint32_t width_, height_;
};

}  // namespace rt_host
