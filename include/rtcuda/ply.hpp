// ply.hpp -- small PLY mesh reader for the render path's drivers (replaces the vendored happly.h of the reference
// for what main.cu:59-61 uses: vertex positions and face index lists).
//
//   rtcuda::PlyMesh mesh("bun_zipper.ply");
//   std::vector<std::array<double, 3>> v_pos   = mesh.getVertexPositions();
//   std::vector<std::vector<size_t>>   f_index = mesh.getFaceIndices<size_t>();
//
// The two accessor names and return types are happly's (happly.h:1451-1465, :1498-1510), so the reference driver
// keeps its lines; define RTCUDA_PLY_AS_HAPPLY before including to also get `happly::PLYData` as an alias.
//
// Formats: ascii, binary_little_endian, binary_big_endian; any scalar property types; list properties with any
// count / item type.  Numeric semantics kept from happly because they decide the fixture geometry bit for bit:
// an ASCII `property float` value is parsed as fp32 (`istream >> float`, happly.h:318-325) and only then widened to
// double; `property double` is parsed as double.  Errors throw std::runtime_error, as happly's do (happly.h:1301).
#ifndef RTCUDA_HOST_PLY_HPP
#define RTCUDA_HOST_PLY_HPP

#include <array>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace rtcuda {

class PlyMesh {
public:
    explicit PlyMesh(const std::string &path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("PLY reader: could not open file " + path);
        parse_header(in);
        if (format_ == ASCII) read_ascii(in);
        else read_binary(in);
    }

    // (x, y, z) of every vertex, widened to double (happly.h:1451-1465)
    std::vector<std::array<double, 3>> getVertexPositions(const std::string &element = "vertex") const {
        const Element &e = find(element);
        const Property &px = e.prop("x"), &py = e.prop("y"), &pz = e.prop("z");
        std::vector<std::array<double, 3>> out(e.count);
        for (size_t i = 0; i < e.count; i++) out[i] = {px.scalars[i], py.scalars[i], pz.scalars[i]};
        return out;
    }

    // the index list of every face (happly.h:1498-1510: element "face", property "vertex_indices" or "vertex_index")
    template <class T>
    std::vector<std::vector<T>> getFaceIndices() const {
        const Element &e = find("face");
        const Property *p = e.maybe("vertex_indices");
        if (!p) p = e.maybe("vertex_index");
        if (!p || !p->is_list) throw std::runtime_error("PLY reader: could not find face vertex indices attribute under any common name.");
        std::vector<std::vector<T>> out(e.count);
        for (size_t i = 0; i < e.count; i++) {
            const size_t lo = p->list_start[i], hi = p->list_start[i + 1];
            out[i].reserve(hi - lo);
            for (size_t k = lo; k < hi; k++) {
                if (p->list_items[k] < 0) throw std::runtime_error("PLY reader: negative face index");
                out[i].push_back((T)p->list_items[k]);
            }
        }
        return out;
    }

    size_t elementCount(const std::string &element) const { return find(element).count; }

private:
    enum Format { ASCII, BINARY_LE, BINARY_BE };
    enum Type { I8, U8, I16, U16, I32, U32, F32, F64 };
    struct Property {
        std::string name;
        bool is_list = false;
        Type type = F32, count_type = U8;
        std::vector<double> scalars;          // scalar property: one value per element entry
        std::vector<long long> list_items;    // list property: all items, entry i = [list_start[i], list_start[i+1])
        std::vector<size_t> list_start;
    };
    struct Element {
        std::string name;
        size_t count = 0;
        std::vector<Property> props;
        const Property *maybe(const std::string &n) const {
            for (const Property &p : props)
                if (p.name == n) return &p;
            return nullptr;
        }
        const Property &prop(const std::string &n) const {
            const Property *p = maybe(n);
            if (!p || p->is_list) throw std::runtime_error("PLY reader: element " + name + " has no scalar property " + n);
            return *p;
        }
    };

    Format format_ = ASCII;
    std::vector<Element> elements_;

    const Element &find(const std::string &n) const {
        for (const Element &e : elements_)
            if (e.name == n) return e;
        throw std::runtime_error("PLY reader: no element named " + n);
    }

    static Type type_of(const std::string &s) {
        static const struct { const char *name; Type t; } table[] = {
            {"char", I8}, {"int8", I8}, {"uchar", U8}, {"uint8", U8}, {"short", I16}, {"int16", I16}, {"ushort", U16},
            {"uint16", U16}, {"int", I32}, {"int32", I32}, {"uint", U32}, {"uint32", U32}, {"float", F32},
            {"float32", F32}, {"double", F64}, {"float64", F64}};
        for (const auto &e : table)
            if (s == e.name) return e.t;
        throw std::runtime_error("PLY reader: unknown property type " + s);
    }
    static size_t size_of(Type t) {
        switch (t) {
            case I8: case U8: return 1;
            case I16: case U16: return 2;
            case I32: case U32: case F32: return 4;
            default: return 8;
        }
    }

    void parse_header(std::istream &in) {
        std::string line;
        if (!std::getline(in, line) || strip(line) != "ply") throw std::runtime_error("PLY reader: not a PLY file (bad magic)");
        bool have_format = false, ended = false;
        while (std::getline(in, line)) {
            std::istringstream ss(strip(line));
            std::string key;
            if (!(ss >> key)) continue;
            if (key == "comment" || key == "obj_info") continue;
            if (key == "format") {
                std::string f, version;
                ss >> f >> version;
                if (f == "ascii") format_ = ASCII;
                else if (f == "binary_little_endian") format_ = BINARY_LE;
                else if (f == "binary_big_endian") format_ = BINARY_BE;
                else throw std::runtime_error("PLY reader: unknown format " + f);
                have_format = true;
            } else if (key == "element") {
                Element e;
                long long n = -1;
                ss >> e.name >> n;
                if (e.name.empty() || n < 0) throw std::runtime_error("PLY reader: malformed element line");
                e.count = (size_t)n;
                elements_.push_back(e);
            } else if (key == "property") {
                if (elements_.empty()) throw std::runtime_error("PLY reader: property before any element");
                Property p;
                std::string t;
                ss >> t;
                if (t == "list") {
                    std::string ct, it;
                    ss >> ct >> it >> p.name;
                    p.is_list = true;
                    p.count_type = type_of(ct);
                    p.type = type_of(it);
                    if (p.count_type == F32 || p.count_type == F64) throw std::runtime_error("PLY reader: list count must be an integer type");
                } else {
                    p.type = type_of(t);
                    ss >> p.name;
                }
                if (p.name.empty()) throw std::runtime_error("PLY reader: malformed property line");
                elements_.back().props.push_back(p);
            } else if (key == "end_header") {
                ended = true;
                break;
            } else {
                throw std::runtime_error("PLY reader: unrecognised header line: " + line);
            }
        }
        if (!have_format || !ended) throw std::runtime_error("PLY reader: incomplete header");
        for (Element &e : elements_)
            for (Property &p : e.props) {
                if (p.is_list) {
                    p.list_start.reserve(e.count + 1);
                    p.list_start.push_back(0);
                } else {
                    p.scalars.reserve(e.count);
                }
            }
    }

    static std::string strip(const std::string &s) {
        size_t a = 0, b = s.size();
        while (b > a && (s[b - 1] == '\r' || s[b - 1] == '\n' || s[b - 1] == ' ' || s[b - 1] == '\t')) b--;
        while (a < b && (s[a] == ' ' || s[a] == '\t')) a++;
        return s.substr(a, b - a);
    }

    // ---- ASCII body: whitespace-separated tokens, element entries in header order
    static double ascii_value(const char *tok, Type t) {
        char *end = nullptr;
        double v;
        // text -> the property's own type first (fp32 for `float`), then double: happly.h:318-325
        if (t == F32) v = (double)strtof(tok, &end);
        else if (t == F64) v = strtod(tok, &end);
        else v = (double)strtoll(tok, &end, 10);
        if (end == tok) throw std::runtime_error(std::string("PLY reader: bad numeric token '") + tok + "'");
        return v;
    }
    void read_ascii(std::istream &in) {
        std::string body((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        size_t pos = 0;
        std::string tok;
        auto next = [&]() -> const char * {
            while (pos < body.size() && (unsigned char)body[pos] <= ' ') pos++;
            if (pos >= body.size()) throw std::runtime_error("PLY reader: unexpected end of file");
            size_t b = pos;
            while (pos < body.size() && (unsigned char)body[pos] > ' ') pos++;
            tok.assign(body, b, pos - b);
            return tok.c_str();
        };
        for (Element &e : elements_)
            for (size_t i = 0; i < e.count; i++)
                for (Property &p : e.props) {
                    if (!p.is_list) {
                        p.scalars.push_back(ascii_value(next(), p.type));
                    } else {
                        long long n = (long long)ascii_value(next(), p.count_type);
                        if (n < 0) throw std::runtime_error("PLY reader: negative list count");
                        for (long long k = 0; k < n; k++) p.list_items.push_back((long long)ascii_value(next(), p.type));
                        p.list_start.push_back(p.list_items.size());
                    }
                }
    }

    // ---- binary body
    double binary_value(std::istream &in, Type t) const {
        unsigned char b[8];
        const size_t n = size_of(t);
        in.read((char *)b, (std::streamsize)n);
        if ((size_t)in.gcount() != n) throw std::runtime_error("PLY reader: unexpected end of file");
        uint64_t u = 0;
        for (size_t k = 0; k < n; k++) u |= (uint64_t)b[format_ == BINARY_LE ? k : n - 1 - k] << (8 * k);
        switch (t) {
            case I8: return (double)(int8_t)u;
            case U8: return (double)(uint8_t)u;
            case I16: return (double)(int16_t)u;
            case U16: return (double)(uint16_t)u;
            case I32: return (double)(int32_t)u;
            case U32: return (double)(uint32_t)u;
            case F32: { uint32_t w = (uint32_t)u; float f; memcpy(&f, &w, 4); return (double)f; }
            default: { double d; memcpy(&d, &u, 8); return d; }
        }
    }
    void read_binary(std::istream &in) {
        for (Element &e : elements_)
            for (size_t i = 0; i < e.count; i++)
                for (Property &p : e.props) {
                    if (!p.is_list) {
                        p.scalars.push_back(binary_value(in, p.type));
                    } else {
                        long long n = (long long)binary_value(in, p.count_type);
                        for (long long k = 0; k < n; k++) p.list_items.push_back((long long)binary_value(in, p.type));
                        p.list_start.push_back(p.list_items.size());
                    }
                }
    }
};

}  // namespace rtcuda

#ifdef RTCUDA_PLY_AS_HAPPLY
namespace happly {
using PLYData = rtcuda::PlyMesh;
}
#endif

#endif  // RTCUDA_HOST_PLY_HPP
