// cornell_bunny.hpp -- the reference driver's scene recipe (main.cu:41-166) as a library function.
//
//   rtcuda::CornellBunny cb("data/bun_zipper.ply", rtcuda::CornellBunny::MATTE);
//   Scene scene = cb.scene();                      // Bvh + lights, ready for render()
//
// MATTE is the reference's scene verbatim: four matte materials (main.cu:42-45), the bunny placed by
// Translate(0.0946899, -0.0329874, -0.0587997) . Scale 2 . Translate(0.3, 0, -0.5) (main.cu:68-70), ten wall
// triangles (:88-107), two light triangles of radiance 15 (:111-116), lights in the order the reference's
// unordered_map iteration yields (:128; last light triangle first -- SURVEY.md Appendix A.13).  The other variants are
// this project's benchmark scenes: FULL_BSDF = glass bunny (ior 1.5) + mirror back wall
// (albedo 0.9); FOUR_BUNNIES = four flattened copies; SIXTEEN_LIGHTS = sixteen 0.1 x 0.1 half-quads of radiance 7.5
// on the ceiling, light order = ascending triangle index.  rtcuda_amd/scenes.py builds the same arrays in numpy;
// tests/test_host_api_cpp.py holds the two against each other bit for bit.
//
// The object owns the host arrays that Primitive / Light point into; keep it alive while its Scene is in use.
#ifndef RTCUDA_HOST_CORNELL_BUNNY_HPP
#define RTCUDA_HOST_CORNELL_BUNNY_HPP

#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "profiler.hpp"
#include "rtcuda.hpp"

namespace rtcuda {

class CornellBunny {
public:
    enum Variant { MATTE, FULL_BSDF, FOUR_BUNNIES, SIXTEEN_LIGHTS };
    enum { RED, GREEN, WHITE, BROWN, GLASS_BUNNY, MIRROR_WALL, NUM_MATERIALS };

    // ply_path empty: the bare box (no bunny)
    // log_stages: print the driver's stage lines ("Reading bunny... done (12.3ms)", main.cu:59-85) through `profiler`
    explicit CornellBunny(const std::string &ply_path, Variant variant = MATTE, bool log_stages_ = false)
        : log_stages(log_stages_), variant_(variant) {
        materials_.resize(NUM_MATERIALS);
        materials_[RED] = Material::make_matte(Vec3(0.65f, 0.05f, 0.05f));
        materials_[GREEN] = Material::make_matte(Vec3(0.12f, 0.45f, 0.15f));
        materials_[WHITE] = Material::make_matte(Vec3(0.73f, 0.73f, 0.73f));
        materials_[BROWN] = Material::make_matte(Vec3(0.62f, 0.57f, 0.54f));
        materials_[GLASS_BUNNY] = Material::make_glass(1.5f);
        materials_[MIRROR_WALL] = Material::make_mirror(Vec3(0.9f, 0.9f, 0.9f));
        if (!ply_path.empty()) add_bunnies(ply_path);
        add_box();
        add_lights();
        // primitives last: they point into vectors that must not grow any more
        primitives_.reserve(triangles_.size());
        for (size_t i = 0; i < triangles_.size(); i++)
            primitives_.emplace_back(&triangles_[i], &materials_[material_of_[i]],
                                     light_of_[i] >= 0 ? &lights_[light_of_[i]] : nullptr);
        bvh_ = Bvh(triangles_, primitives_);
    }
    CornellBunny(const CornellBunny &) = delete;
    CornellBunny &operator=(const CornellBunny &) = delete;

    Scene scene() { return Scene{bvh_, (int)lights_.size(), lights_.data()}; }
    // main.cu:162-166
    static Camera camera(float aspect_ratio) {
        return Camera(Vec3(0.5f, 0.5f, 1.5f), Vec3(0.5f, 0.5f, 0.0f), Vec3(0.0f, 1.0f, 0.0f), 37.8f, aspect_ratio);
    }

    const std::vector<Triangle> &triangles() const { return triangles_; }
    const std::vector<Material> &materials() const { return materials_; }
    const std::vector<Light> &lights() const { return lights_; }
    const std::vector<int> &material_of() const { return material_of_; }  // per triangle: index into materials()
    const std::vector<int> &light_of() const { return light_of_; }        // per triangle: index into lights() or -1
    size_t num_vertices = 0, num_faces = 0;                               // of the PLY (0 for the bare box)
    const bool log_stages;

private:
    void add_triangle(const Vec3 &a, const Vec3 &b, const Vec3 &c, int material) {
        triangles_.emplace_back(a, b, c);
        material_of_.push_back(material);
        light_of_.push_back(-1);
    }

    void add_bunnies(const std::string &ply_path) {
        // (stage lines as the reference's driver prints them -- main.cu:59-85 -- when `log_stages` is set)
        if (log_stages) profiler.start("Reading bunny");
        PlyMesh mesh(ply_path);
        const std::vector<std::array<double, 3>> rest = mesh.getVertexPositions();
        const std::vector<std::vector<size_t>> faces = mesh.getFaceIndices<size_t>();
        num_vertices = rest.size();
        num_faces = faces.size();
        if (log_stages) profiler.stop();
        if (log_stages) std::cout << num_vertices << " vertices, " << num_faces << " faces" << std::endl;
        static const float one[1][3] = {{0.3f, 0.f, -0.5f}};
        static const float four[4][3] = {{0.1f, 0.f, -0.3f}, {0.55f, 0.f, -0.3f}, {0.1f, 0.f, -0.62f}, {0.55f, 0.f, -0.62f}};
        const float(*places)[3] = variant_ == FOUR_BUNNIES ? four : one;
        const int copies = variant_ == FOUR_BUNNIES ? 4 : 1;
        const int material = variant_ == FULL_BSDF ? GLASS_BUNNY : BROWN;
        if (log_stages) profiler.start("Transforming bunny");
        std::vector<std::vector<std::array<double, 3>>> placed((size_t)copies, rest);
        for (int c = 0; c < copies; c++) {
            Transform t(Matrix4x4::Translate(0.0946899f, -0.0329874f, -0.0587997f));
            t.composite(Matrix4x4::Scale(2.f, 2.f, 2.f));
            t.composite(Matrix4x4::Translate(places[c][0], places[c][1], places[c][2]));
            for (auto &p : placed[(size_t)c]) t.apply(p);
        }
        if (log_stages) profiler.stop();
        Profiler::Stage converting(log_stages ? profiler : quiet_, "Converting bunny to triangles");
        for (int c = 0; c < copies; c++) {
            const std::vector<std::array<double, 3>> &v = placed[(size_t)c];
            for (const std::vector<size_t> &f : faces) {
                if (f.size() != 3) throw std::runtime_error("CornellBunny: the mesh has a face that is not a triangle");
                for (size_t k : f)
                    if (k >= v.size()) throw std::runtime_error("CornellBunny: face index out of range");
                // the double -> float narrowing of each coordinate happens here, as at main.cu:79-81
                add_triangle(Vec3((float)v[f[0]][0], (float)v[f[0]][1], (float)v[f[0]][2]),
                             Vec3((float)v[f[1]][0], (float)v[f[1]][1], (float)v[f[1]][2]),
                             Vec3((float)v[f[2]][0], (float)v[f[2]][1], (float)v[f[2]][2]), material);
            }
        }
    }

    // the unit box x, y in [0, 1], z in [-1, 0], open towards the camera: two triangles per wall, each given by
    // the corner the two share first and the diagonally opposite one last
    void add_box() {
        struct Wall { float a[3], b[3], c[3], d[3]; int material; };  // triangles (a, b, d) and (a, c, d)
        const int back = variant_ == FULL_BSDF ? MIRROR_WALL : WHITE;
        const Wall walls[5] = {
            {{0, 0, 0}, {0, 0, -1}, {0, 1, 0}, {0, 1, -1}, RED},      // left    main.cu:88-91
            {{1, 0, 0}, {1, 0, -1}, {1, 1, 0}, {1, 1, -1}, GREEN},    // right   :92-95
            {{0, 0, 0}, {1, 0, 0}, {0, 0, -1}, {1, 0, -1}, WHITE},    // floor   :96-99
            {{0, 1, 0}, {1, 1, 0}, {0, 1, -1}, {1, 1, -1}, WHITE},    // ceiling :100-103
            {{0, 0, -1}, {1, 0, -1}, {0, 1, -1}, {1, 1, -1}, back},   // back    :104-107
        };
        auto v = [](const float *p) { return Vec3(p[0], p[1], p[2]); };
        for (const Wall &w : walls) {
            add_triangle(v(w.a), v(w.b), v(w.d), w.material);
            add_triangle(v(w.a), v(w.c), v(w.d), w.material);
        }
    }

    void add_lights() {
        const float y = 0.999f;
        const size_t first = triangles_.size();
        float radiance = 15.f;
        auto quad = [&](float x0, float x1, float z0, float z1) {  // z0 is the edge nearer the camera
            add_triangle(Vec3(x0, y, z0), Vec3(x1, y, z0), Vec3(x1, y, z1), WHITE);
            add_triangle(Vec3(x0, y, z0), Vec3(x0, y, z1), Vec3(x1, y, z1), WHITE);
        };
        if (variant_ == SIXTEEN_LIGHTS) {
            radiance = 7.5f;
            const float half = 0.05f, zs[2] = {-0.35f, -0.65f}, xs[4] = {0.2f, 0.4f, 0.6f, 0.8f};
            for (float cz : zs)
                for (float cx : xs) quad(cx - half, cx + half, cz + half, cz - half);  // corners in fp32
        } else {
            quad(0.4f, 0.6f, -0.4f, -0.6f);  // main.cu:111-116
        }
        const size_t n = triangles_.size() - first;
        lights_.reserve(n);
        for (size_t k = 0; k < n; k++) {
            // the reference scene's order is its unordered_map's: the LAST light triangle is light 0
            const size_t tri = variant_ == SIXTEEN_LIGHTS ? first + k : triangles_.size() - 1 - k;
            lights_.push_back(Light::make_area_light(&triangles_[tri], Vec3(radiance, radiance, radiance)));
            light_of_[tri] = (int)k;
        }
    }

    Variant variant_;
    Profiler quiet_{false};  // (a disabled profiler: Stage guards that print nothing)
    std::vector<Material> materials_;
    std::vector<Triangle> triangles_;
    std::vector<int> material_of_, light_of_;
    std::vector<Light> lights_;
    std::vector<Primitive> primitives_;
    Bvh bvh_;
};

// The driver's image output (main.cu:178-191): P3 text, clamp(int(256 c), 0, 255), row 0 first.
inline void write_ppm(const std::string &path, int width, int height, const std::vector<Vec3> &framebuffer) {
    std::ofstream file(path);
    if (!file) throw std::runtime_error("write_ppm: cannot open " + path);
    auto q = [](float c) { int v = int(256.f * c); return v < 0 ? 0 : (v > 255 ? 255 : v); };
    file << "P3\n" << width << ' ' << height << "\n255\n";
    for (size_t k = 0; k < (size_t)width * height; k++)
        file << q(framebuffer[k].x) << ' ' << q(framebuffer[k].y) << ' ' << q(framebuffer[k].z) << '\n';
}

}  // namespace rtcuda

#endif  // RTCUDA_HOST_CORNELL_BUNNY_HPP
