// Exercises include/rt_host.hpp the way the reference's main loop uses its renderer:
// build the pyramid scene (src/scenes/pyramid.rs:22-61), construct the tracer, render one frame from the
// scene's start camera (src/camera.rs:64) and dump the hit records. Exit code 2 = no usable GPU.
#include <cstdio>
#include <cstring>

#include "rt_host.hpp"

static TriangleData tri(const float* a, const float* b, const float* c, uint32_t material) {
    TriangleData t;
    memset(&t, 0, sizeof(t));
    memcpy(t.v0, a, 12);
    memcpy(t.v1, b, 12);
    memcpy(t.v2, c, 12);
    t.material_id = (float)material;
    return t;
}

int main(int argc, char** argv) {
    const float apex[3] = {0, 5, 0}, p0[3] = {-4, 0, -4}, p1[3] = {4, 0, -4}, p2[3] = {4, 0, 4}, p3[3] = {-4, 0, 4};
    std::vector<TriangleData> tris = {tri(p0, p1, apex, 0), tri(p1, p2, apex, 1), tri(p2, p3, apex, 2),
                                      tri(p3, p0, apex, 3), tri(p0, p2, p1, 4),   tri(p0, p3, p2, 4)};
    try {
        rt_host::RayTracer tracer({}, tris, 64, 36);
        rt_host::Frame f = tracer.render(rt_host::Camera::for_scene("pyramid"), 0.0f);
        size_t hits = 0;
        for (const rt_hit& h : f.primary) hits += h.id != RT_MISS;
        printf("hits %zu\n", hits);
        if (argc > 1) {
            FILE* fp = fopen(argv[1], "wb");
            if (!fp) return 3;
            fwrite(f.primary.data(), sizeof(rt_hit), f.primary.size(), fp);
            fwrite(f.shadow.data(), sizeof(rt_hit), f.shadow.size(), fp);
            // the shaded picture, as ComputeState::render hands it out
            const std::vector<uint8_t> rgba = tracer.render_rgba(rt_host::Camera::for_scene("pyramid"), 0.0f);
            fwrite(rgba.data(), 1, rgba.size(), fp);
            fclose(fp);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 2;
    }
    return 0;
}
