// cornell_bunny.cpp -- the reference driver's scene (main.cu:39-194) against the drop-in host API.
//
// Same recipe: four matte materials, bun_zipper.ply transformed by Translate . Scale . Translate,
// ten wall triangles, two light triangles, Bvh, Scene, Camera, render(), image.ppm.  What differs
// from main.cu: no cudaMalloc/cudaMemcpy (the pointers are host pointers), an own small PLY reader
// instead of happly, and command-line width/height/spp/output (main.cu hard-codes 600x600x10).
//
//   hipcc -O2 -std=c++17 -I include examples/cornell_bunny.cpp -L rtcuda_amd -lrtcuda_amd \
//         -Wl,-rpath,'$ORIGIN/../rtcuda_amd' -o examples/cornell_bunny     (make -C rtcuda_amd/csrc example)
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>

#include "rtcuda/rtcuda.hpp"

// ---- matrix4x4.hpp:22-34 / transform.hpp:13-33 (host scene preparation, fp32 composite, mixed apply)
struct Matrix4x4 {
    float data[4][4];
    static Matrix4x4 Identity() { Matrix4x4 m{}; for (int i = 0; i < 4; i++) m.data[i][i] = 1.f; return m; }
    static Matrix4x4 Translate(float dx, float dy, float dz) { Matrix4x4 m = Identity(); m.data[0][3] = dx; m.data[1][3] = dy; m.data[2][3] = dz; return m; }
    static Matrix4x4 Scale(float sx, float sy, float sz) { Matrix4x4 m = Identity(); m.data[0][0] = sx; m.data[1][1] = sy; m.data[2][2] = sz; return m; }
};
struct Transform {
    explicit Transform(const Matrix4x4 &m) : matrix(m) {}
    void composite(const Matrix4x4 &other) {  // result = other . matrix, fp32
        Matrix4x4 r;
        for (int i = 0; i < 4; i++)
            for (int j = 0; j < 4; j++) {
                r.data[i][j] = 0;
                for (int k = 0; k < 4; ++k) r.data[i][j] += other.data[i][k] * matrix.data[k][j];
            }
        matrix = r;
    }
    void apply(std::array<double, 3> &v) const {  // x, y rounded to float; z stays double
        float nx = matrix.data[0][0] * v[0] + matrix.data[0][1] * v[1] + matrix.data[0][2] * v[2] + matrix.data[0][3];
        float ny = matrix.data[1][0] * v[0] + matrix.data[1][1] * v[1] + matrix.data[1][2] * v[2] + matrix.data[1][3];
        v[2] = matrix.data[2][0] * v[0] + matrix.data[2][1] * v[1] + matrix.data[2][2] * v[2] + matrix.data[2][3];
        v[0] = nx;
        v[1] = ny;
    }
    Matrix4x4 matrix;
};

// ---- minimal ASCII PLY reader: property float x y z first, triangle faces (happly.h:318-325 semantics:
//      text -> float -> double)
static bool load_ply(const std::string &path, std::vector<std::array<double, 3>> &v_pos,
                     std::vector<std::array<size_t, 3>> &faces) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    size_t nv = 0, nf = 0, nprops = 0;
    bool in_vertex = false;
    while (std::getline(in, line)) {
        std::istringstream ss(line);
        std::string tok;
        ss >> tok;
        if (tok == "element") { std::string name; size_t cnt; ss >> name >> cnt; in_vertex = name == "vertex"; if (in_vertex) nv = cnt; else if (name == "face") nf = cnt; }
        else if (tok == "property" && in_vertex) nprops++;
        else if (tok == "end_header") break;
    }
    v_pos.resize(nv);
    for (size_t i = 0; i < nv; i++) {
        std::getline(in, line);
        std::istringstream ss(line);
        for (size_t k = 0; k < nprops; k++) { float f; ss >> f; if (k < 3) v_pos[i][k] = f; }
    }
    faces.resize(nf);
    for (size_t i = 0; i < nf; i++) {
        size_t cnt;
        in >> cnt >> faces[i][0] >> faces[i][1] >> faces[i][2];
        if (cnt != 3) return false;
    }
    return true;
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

int main(int argc, char **argv) {
    const int WIDTH = argc > 1 ? atoi(argv[1]) : 600, HEIGHT = argc > 2 ? atoi(argv[2]) : 600;
    const int NUM_SAMPLES = argc > 3 ? atoi(argv[3]) : 10;
    const std::string ply = argc > 4 ? argv[4] : "data/bun_zipper.ply";
    const std::string out = argc > 5 ? argv[5] : "image.ppm";
    try {
        std::vector<Material> materials;  // main.cu:41-56 (host storage instead of cudaMalloc)
        materials.push_back(Material::make_matte(Vec3(0.65f, 0.05f, 0.05f)));
        materials.push_back(Material::make_matte(Vec3(0.12f, 0.45f, 0.15f)));
        materials.push_back(Material::make_matte(Vec3(0.73f, 0.73f, 0.73f)));
        materials.push_back(Material::make_matte(Vec3(0.62f, 0.57f, 0.54f)));
        Material *d_red = &materials[0], *d_green = &materials[1], *d_white = &materials[2], *d_brown = &materials[3];

        std::vector<std::array<double, 3>> v_pos;
        std::vector<std::array<size_t, 3>> f_index;
        if (!load_ply(ply, v_pos, f_index)) throw std::runtime_error("cannot read " + ply);
        std::cout << v_pos.size() << " vertices, " << f_index.size() << " faces" << std::endl;

        Transform transform(Matrix4x4::Translate(0.0946899f, -0.0329874f, -0.0587997f));  // main.cu:68-71
        transform.composite(Matrix4x4::Scale(2.f, 2.f, 2.f));
        transform.composite(Matrix4x4::Translate(0.3f, 0.f, -0.5f));
        for (auto &v : v_pos) transform.apply(v);

        std::vector<Triangle> triangles;
        std::vector<Material *> material_ptrs;
        for (auto &face : f_index) {
            triangles.emplace_back(Vec3(v_pos[face[0]][0], v_pos[face[0]][1], v_pos[face[0]][2]),
                                   Vec3(v_pos[face[1]][0], v_pos[face[1]][1], v_pos[face[1]][2]),
                                   Vec3(v_pos[face[2]][0], v_pos[face[2]][1], v_pos[face[2]][2]));
            material_ptrs.push_back(d_brown);
        }
        auto wall = [&](Vec3 a, Vec3 b, Vec3 c, Material *m) { triangles.emplace_back(a, b, c); material_ptrs.push_back(m); };
        wall(Vec3(0, 0, 0), Vec3(0, 0, -1), Vec3(0, 1, -1), d_red);     // main.cu:88-107
        wall(Vec3(0, 0, 0), Vec3(0, 1, 0), Vec3(0, 1, -1), d_red);
        wall(Vec3(1, 0, 0), Vec3(1, 0, -1), Vec3(1, 1, -1), d_green);
        wall(Vec3(1, 0, 0), Vec3(1, 1, 0), Vec3(1, 1, -1), d_green);
        wall(Vec3(0, 0, 0), Vec3(1, 0, 0), Vec3(1, 0, -1), d_white);
        wall(Vec3(0, 0, 0), Vec3(0, 0, -1), Vec3(1, 0, -1), d_white);
        wall(Vec3(0, 1, 0), Vec3(1, 1, 0), Vec3(1, 1, -1), d_white);
        wall(Vec3(0, 1, 0), Vec3(0, 1, -1), Vec3(1, 1, -1), d_white);
        wall(Vec3(0, 0, -1), Vec3(1, 0, -1), Vec3(1, 1, -1), d_white);
        wall(Vec3(0, 0, -1), Vec3(0, 1, -1), Vec3(1, 1, -1), d_white);
        wall(Vec3(0.4f, 0.999f, -0.4f), Vec3(0.6f, 0.999f, -0.4f), Vec3(0.6f, 0.999f, -0.6f), d_white);  // :111-116
        wall(Vec3(0.4f, 0.999f, -0.4f), Vec3(0.4f, 0.999f, -0.6f), Vec3(0.6f, 0.999f, -0.6f), d_white);
        const int num_triangles = (int)triangles.size();

        // lights in the order the reference's unordered_map iteration yields (main.cu:128): last triangle first
        std::vector<Light> lights;
        lights.push_back(Light::make_area_light(&triangles[num_triangles - 1], Vec3(15.f, 15.f, 15.f)));
        lights.push_back(Light::make_area_light(&triangles[num_triangles - 2], Vec3(15.f, 15.f, 15.f)));

        std::vector<Primitive> primitives;  // main.cu:141-148
        for (int i = 0; i < num_triangles; i++) {
            if (i == num_triangles - 1) primitives.emplace_back(&triangles[i], material_ptrs[i], &lights[0]);
            else if (i == num_triangles - 2) primitives.emplace_back(&triangles[i], material_ptrs[i], &lights[1]);
            else primitives.emplace_back(&triangles[i], material_ptrs[i]);
        }
        Bvh bvh(triangles, primitives);  // main.cu:151
        Scene scene = {bvh, (int)lights.size(), lights.data()};

        Camera camera(Vec3(0.5f, 0.5f, 1.5f), Vec3(0.5f, 0.5f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 37.8f,
                      (float)WIDTH / (float)HEIGHT);
        std::vector<Vec3> framebuffer;
        rt_stats st;
        auto t0 = std::chrono::steady_clock::now();
        render(WIDTH, HEIGHT, NUM_SAMPLES, 10, camera, scene, framebuffer, 1, &st);
        float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Rendering... done (" << ms << "ms), render loop " << st.seconds_render * 1e3 << " ms, "
                  << (double)WIDTH * HEIGHT * NUM_SAMPLES / st.seconds_render / 1e6 << " Msamples/s" << std::endl;

        std::ofstream file(out);  // main.cu:178-191
        file << "P3\n" << WIDTH << ' ' << HEIGHT << "\n255\n";
        for (int j = 0; j < HEIGHT; j++)
            for (int i = 0; i < WIDTH; i++) {
                const Vec3 &c = framebuffer[(size_t)j * WIDTH + i];
                file << clampi(int(256.f * c.x), 0, 255) << ' ' << clampi(int(256.f * c.y), 0, 255) << ' '
                     << clampi(int(256.f * c.z), 0, 255) << "\n";
            }
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
