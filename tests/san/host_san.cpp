// The C++ scene API above the C ABI under AddressSanitizer / UBSan: the built-in scenes, the scene.txt loader on the
// sample file, on every prefix of it (a file cut short anywhere must be refused or loaded, never crash or hang) and on
// copies with one line dropped or one token garbled, the two-scenes-alive case and the creation-order check.
// Compiled by tests/test_host_sanitizers.py as host_capi.cpp + this file against the built library; nothing here
// touches a GPU (no upload, no render).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

extern "C" {
const char *pth_error();
void pth_close();
int pth_open_builtin(int which);
int pth_open_file(const char *path, int honor_center, int *width, int *height, int *spp);
int pth_counts(int *n_prims, int *n_materials);
int pth_copy(uint8_t *prim_type, float *prim_param, uint32_t *prim_material, uint8_t *mat_type, float *mat_param, pt_camera *cam, float *background);
int pth_two_scenes_alive();
int pth_ordering_violation();
}

static int copy_out()
{
    int n = 0, m = 0;
    if (pth_counts(&n, &m) != 0) return 1;
    std::vector<uint8_t> pt((size_t)n), mt((size_t)m);
    std::vector<float> pp((size_t)n * 6), mp((size_t)m * 4);
    std::vector<uint32_t> pm((size_t)n);
    pt_camera cam;
    float bg[3];
    return pth_copy(pt.data(), pp.data(), pm.data(), mt.data(), mp.data(), &cam, bg);
}

static void write_file(const std::string &path, const std::string &text)
{
    std::ofstream f(path, std::ios::binary);
    f << text;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_san <sample scene.txt> <scratch dir>\n"); return 2; }
    const std::string sample = argv[1], tmp = std::string(argv[2]) + "/host_san_scene.txt";
    for (int which = 0; which < 2; ++which) {
        if (pth_open_builtin(which) != 0) { fprintf(stderr, "host_san: builtin %d: %s\n", which, pth_error()); return 1; }
        if (copy_out() != 0) return 1;
    }
    int w = 0, h = 0, s = 0;
    for (int honor = 0; honor < 2; ++honor) {
        if (pth_open_file(sample.c_str(), honor, &w, &h, &s) != 0) { fprintf(stderr, "host_san: sample: %s\n", pth_error()); return 1; }
        if (copy_out() != 0) return 1;
    }
    std::ifstream in(sample, std::ios::binary);
    std::stringstream ss;
    ss << in.rdbuf();
    const std::string text = ss.str();
    int loaded = 0, refused = 0;
    // every prefix of the file (stepping by 3 bytes keeps this under a second)
    for (size_t cut = 0; cut < text.size(); cut += 3) {
        write_file(tmp, text.substr(0, cut));
        if (pth_open_file(tmp.c_str(), 0, &w, &h, &s) == 0) { ++loaded; if (copy_out() != 0) return 1; } else ++refused;
    }
    // one line dropped; one '=' value replaced by something that is not a number
    std::vector<std::string> lines;
    {
        std::stringstream ls(text);
        std::string l;
        while (std::getline(ls, l)) lines.push_back(l);
    }
    for (size_t drop = 0; drop < lines.size(); ++drop) {
        std::string t;
        for (size_t k = 0; k < lines.size(); ++k)
            if (k != drop) t += lines[k] + "\n";
        write_file(tmp, t);
        if (pth_open_file(tmp.c_str(), 1, &w, &h, &s) == 0) { ++loaded; if (copy_out() != 0) return 1; } else ++refused;
    }
    for (size_t garble = 0; garble < lines.size(); ++garble) {
        const size_t eq = lines[garble].find('=');
        if (eq == std::string::npos) continue;
        for (const char *junk : {"", " ", "x", "-", "1e999", "nan", "99999999999999999999", "1 2 3 4 5 6 7 8 9"}) {
            std::string t;
            for (size_t k = 0; k < lines.size(); ++k) t += (k == garble ? lines[k].substr(0, eq + 1) + junk : lines[k]) + "\n";
            write_file(tmp, t);
            if (pth_open_file(tmp.c_str(), 0, &w, &h, &s) == 0) { ++loaded; if (copy_out() != 0) return 1; } else ++refused;
        }
    }
    std::remove(tmp.c_str());
    if (pth_two_scenes_alive() != 1) { fprintf(stderr, "host_san: two scenes alive: %s\n", pth_error()); return 1; }
    if (pth_ordering_violation() != 1) { fprintf(stderr, "host_san: ordering violation not refused\n"); return 1; }
    pth_close();
    printf("ok %d variants loaded, %d refused\n", loaded, refused);
    return 0;
}
