// cornell.cpp — the Cornell-box scene of rpt's examples/cornell.rs written against the C++
// mirror of the builder API (include/rpt.hpp).  Usage: cornell [size] [spp] [out.ppm]
// Prints one line: "<width> <height> <spp> <fnv1a-64 of the RGB bytes> <variance>".
#include <cstdio>
#include <cstdlib>
#include <string>

#include "rpt.hpp"

using namespace rpt;

int main(int argc, char** argv) {
    uint32_t size = argc > 1 ? uint32_t(std::atoi(argv[1])) : 512;
    uint32_t spp = argc > 2 ? uint32_t(std::atoi(argv[2])) : 64;
    std::string out = argc > 3 ? argv[3] : "";
    try {
        Scene scene = Scene::new_();
        Camera camera;
        camera.eye = vec3(278.0, 273.0, -800.0);
        camera.direction = vec3(0.0, 0.0, 1.0);
        camera.up = vec3(0.0, 1.0, 0.0);
        camera.fov = 0.686;

        Material white = Material::diffuse(hex_color(0xAAAAAA));
        Material red = Material::diffuse(hex_color(0xBC0000));
        Material green = Material::diffuse(hex_color(0x00BC00));
        Material light_mtl = Material::light(hex_color(0xFFFEFA), 100.0);

        Shape floor = polygon({vec3(0.0, 0.0, 0.0), vec3(0.0, 0.0, 559.2), vec3(556.0, 0.0, 559.2), vec3(556.0, 0.0, 0.0)});
        Shape ceiling = polygon({vec3(0.0, 548.9, 0.0), vec3(556.0, 548.9, 0.0), vec3(556.0, 548.9, 559.2), vec3(0.0, 548.9, 559.2)});
        Shape light_rect = polygon({vec3(343.0, 548.8, 227.0), vec3(343.0, 548.8, 332.0), vec3(213.0, 548.8, 332.0), vec3(213.0, 548.8, 227.0)});
        Shape back_wall = polygon({vec3(0.0, 0.0, 559.2), vec3(0.0, 548.9, 559.2), vec3(556.0, 548.9, 559.2), vec3(556.0, 0.0, 559.2)});
        Shape right_wall = polygon({vec3(0.0, 0.0, 0.0), vec3(0.0, 548.9, 0.0), vec3(0.0, 548.9, 559.2), vec3(0.0, 0.0, 559.2)});
        Shape left_wall = polygon({vec3(556.0, 0.0, 0.0), vec3(556.0, 0.0, 559.2), vec3(556.0, 548.9, 559.2), vec3(556.0, 548.9, 0.0)});
        const double two_pi = 6.283185307179586476925286766559;
        Shape large_box = cube().scale(vec3(165.0, 330.0, 165.0)).rotate_y(two_pi * (-253.0 / 360.0)).translate(vec3(368.0, 165.0, 351.0));
        Shape small_box = sphere().scale(vec3(80.0, 80.0, 80.0)).rotate_y(two_pi * (-197.0 / 360.0)).translate(vec3(150.0, 82.5, 450.0));

        scene.add(Object(floor).material(white));
        scene.add(Object(ceiling).material(white));
        scene.add(Object(back_wall).material(white));
        scene.add(Object(left_wall).material(red));
        scene.add(Object(right_wall).material(green));
        scene.add(Object(large_box).material(white));
        scene.add(Object(small_box).material(white));
        scene.add(std::make_pair(light_rect, light_mtl));  // light and object at the same time

        RgbImage last;
        double variance = 0.0;
        uint32_t interval = spp >= 2 ? spp / 2 : spp;
        Renderer(scene, camera)
            .width(size)
            .height(size)
            .filter(Filter::Box(1))
            .max_bounces(2)
            .num_samples(spp)
            .seed(1)
            .iterative_render(interval, [&](uint32_t iteration, const Buffer& buffer) {
                last = buffer.image();
                if (buffer.batches() > 1) variance = buffer.variance();
                std::fprintf(stderr, "Finished iteration %u\n", iteration);
            });
        if (argc > 4) {  // optional: photon-mapped render of the same scene (photon.rs:650-652 style entry point)
            Renderer pr(scene, camera);
            pr.width(size).height(size).num_samples(2).gather_size(20).gather_size_volume(3).watts(14.65 * 5000.0).seed(1);
            RgbImage pimg = pr.photon_point_query_beam_render(5000);
            size_t lit = 0;
            for (uint8_t b : pimg.data) lit += b > 0;
            std::fprintf(stderr, "photon render: %zu non-zero bytes\n", lit);
            if (lit == 0) return 3;
        }
        uint64_t h = 1469598103934665603ull;
        for (uint8_t b : last.data) {
            h ^= b;
            h *= 1099511628211ull;
        }
        std::printf("%u %u %u %016llx %.17g\n", last.width, last.height, spp, (unsigned long long)h, variance);
        if (!out.empty()) {
            FILE* f = std::fopen(out.c_str(), "wb");
            if (!f) return 2;
            std::fprintf(f, "P6\n%u %u\n255\n", last.width, last.height);
            std::fwrite(last.data.data(), 1, last.data.size(), f);
            std::fclose(f);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
