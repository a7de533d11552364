// main.cpp -- the program of the reference (main.cu:106-220) on the MI355X path: reads the optional
// scene.txt from the working directory, renders, writes image.bmp, prints the same stderr lines.
//   pathtracer [--depth N] [--seed S] [--spp N] [--pfm file] [--scene N] [--gpus N]   (flags are additions)
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include "render.h"
#include "scene.h"

int main(int argc, char **argv)
{
    std::ifstream file;
    file.open("scene.txt");
    float aspect_ratio = 1.33f;
    int image_width = 1200;
    int image_height = static_cast<int>(image_width / aspect_ratio);
    int tx = 8, ty = 8; // thread-block side of the reference's launch; kept for the file format, unused here
    int samples_per_pixel = 80;
    int builtin = 0, spp_flag = -1;
    std::string pfm;
    for (int a = 1; a < argc; ++a) {
        auto val = [&](const char *flag) { if (a + 1 >= argc) { std::cerr << flag << " needs a value\n"; exit(2); } return argv[++a]; };
        if (!strcmp(argv[a], "--depth")) pt_host::options().max_depth = atoi(val("--depth"));
        else if (!strcmp(argv[a], "--seed")) pt_host::options().seed = strtoull(val("--seed"), nullptr, 10);
        else if (!strcmp(argv[a], "--spp")) spp_flag = atoi(val("--spp"));
        else if (!strcmp(argv[a], "--pfm")) pfm = val("--pfm");
        else if (!strcmp(argv[a], "--scene")) builtin = atoi(val("--scene"));
        else if (!strcmp(argv[a], "--gpus")) pt_host::options().gpus = atoi(val("--gpus")); // the samples split over GPUs 0..N-1, one RCCL all-reduce
        else { std::cerr << "unknown flag " << argv[a] << "\n"; return 2; }
    }
    scene *sc = nullptr;
    try {
        if (!file.is_open()) {
            sc = builtin ? new scene(builtin) : new scene();
        } else {
            std::string line;
            do {
                if (!std::getline(file, line)) { line = "END"; break; } // the reference loops forever here
                if (!line.empty() && line.back() == '\r') line.pop_back();
                if (line == "Image") continue;
                auto kv = get_key_value(line);
                if (kv[0] == "width") image_width = std::stoi(kv[1]);
                else if (kv[0] == "height") image_height = std::stoi(kv[1]);
                else if (kv[0] == "square_side") tx = ty = std::stoi(kv[1]);
                else if (kv[0] == "samples_per_pixel") samples_per_pixel = std::stoi(kv[1]);
            } while (line != "Scene" && line != "END");
            aspect_ratio = (float)image_width / (float)image_height;
            if (line != "Scene") {
                std::perror("Wrong file format");
                return -1;
            }
            sc = new scene(file, aspect_ratio);
        }
        if (spp_flag > 0) samples_per_pixel = spp_flag;
        std::cerr << "Rendering a " << image_width << "x" << image_height << " image ";
        std::cerr << "in " << tx << "x" << ty << " blocks.\n";

        std::vector<vec3> fb((size_t)image_width * image_height);
        auto t0 = std::chrono::steady_clock::now();
        pt_stats st;
        pt_host::render(fb.data(), image_width, image_height, samples_per_pixel, sc->background_color, sc->cam->d_this, sc->world->d_this, &st);
        auto t1 = std::chrono::steady_clock::now();
        std::cerr << "Cuda computation took " << std::chrono::duration<double>(t1 - t0).count() << " seconds.\n";
        std::cerr << st.rays << " rays, " << st.rays / st.ms_total / 1e3 << " Mrays/s on the device.\n";

        auto pixels = pt_host::quantise(fb.data(), image_width, image_height);
        bool ok = pt_host::write_bmp("image.bmp", image_width, image_height, pixels.data());
        if (ok && !pfm.empty()) ok = pt_host::write_pfm(pfm, image_width, image_height, fb.data());
        auto t2 = std::chrono::steady_clock::now();
        std::cerr << "Image write took " << std::chrono::duration<double>(t2 - t1).count() << " seconds. \n";
        delete sc;
        return ok ? 0 : 1;
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        delete sc;
        return 99; // the exit code of the reference's checkCudaErrors, reached without killing the process mid-call
    }
}
