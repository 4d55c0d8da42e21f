// Wavefront OBJ reader on the host (C++), with upstream's semantics (src/mesh/mesh.py:211-251),
// which define object and face numbering:
//   * only "o ", "v " and "f " records are read (after stripping the line);
//   * an "o" record closes the previous object: its vertices and faces are the ones collected since
//     the previous "o" (records before the first "o" end up in the first object), and face indices
//     are rebased by the vertex count of all EARLIER closed objects;
//   * a face corner a/b/c contributes its first index; polygons are fan-triangulated around their
//     first corner; faces with fewer than three corners are dropped;
//   * a file without any "o" record yields no object.
// Numbers are parsed with strtod / strtol (correctly rounded, like Python's float()).
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rt {

struct ObjFile {
    std::vector<std::string> names;
    std::vector<int64_t> vertex_begin, tri_begin;   // per object, plus one-past-the-end
    std::vector<double> vertices;                   // [n][3]
    std::vector<int32_t> triangles;                 // [m][3], indices local to the object
    std::string error;
};

inline bool parse_obj(const char *path, ObjFile &out) {
    FILE *fh = std::fopen(path, "rb");
    if (!fh) { out.error = std::string("cannot open ") + path; return false; }
    std::string data;
    char buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof buf, fh)) > 0) data.append(buf, got);
    std::fclose(fh);

    int64_t offset = 0;              // vertices of closed objects, subtracted from face indices
    int64_t open_vertices = 0;       // vertices collected since the last close
    bool any_object = false;
    std::vector<int64_t> corner;
    out.vertex_begin.push_back(0); out.tri_begin.push_back(0);
    const char *p = data.c_str(), *end = p + data.size();
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; };
    while (p < end) {
        const char *eol = static_cast<const char *>(std::memchr(p, '\n', (size_t)(end - p)));
        if (!eol) eol = end;
        const char *a = p, *b = eol;
        while (a < b && is_space(*a)) ++a;              // line.strip()
        while (b > a && is_space(b[-1])) --b;
        if (b - a >= 2 && a[1] == ' ') {
            if (a[0] == 'o') {
                if (any_object) {                        // close the previous object
                    out.vertex_begin.push_back((int64_t)out.vertices.size() / 3);
                    out.tri_begin.push_back((int64_t)out.triangles.size() / 3);
                    offset += open_vertices;
                    open_vertices = 0;
                }
                any_object = true;
                out.names.emplace_back(a + 2, (size_t)(b - a - 2));   // line.split(" ", 1)[1]
            } else if (a[0] == 'v') {
                const std::string line(a + 2, (size_t)(b - a - 2));
                const char *q = line.c_str();
                char *next = nullptr;
                for (int k = 0; k < 3; ++k) {
                    const double v = std::strtod(q, &next);
                    if (next == q) { out.error = "bad vertex record: v " + line; return false; }
                    out.vertices.push_back(v);
                    q = next;
                }
                ++open_vertices;
            } else if (a[0] == 'f') {
                const std::string line(a + 2, (size_t)(b - a - 2));
                corner.clear();
                const char *q = line.c_str();
                while (*q) {
                    while (*q && is_space(*q)) ++q;
                    if (!*q) break;
                    char *next = nullptr;
                    const long long idx = std::strtoll(q, &next, 10);   // the part before the first '/'
                    if (next == q) { out.error = "bad face record: f " + line; return false; }
                    corner.push_back((int64_t)idx - 1 - offset);
                    q = next;
                    while (*q && !is_space(*q)) ++q;                    // skip /b/c
                }
                for (size_t k = 1; k + 1 < corner.size(); ++k) {
                    out.triangles.push_back((int32_t)corner[0]);
                    out.triangles.push_back((int32_t)corner[k]);
                    out.triangles.push_back((int32_t)corner[k + 1]);
                }
            }
        }
        p = eol + 1;
    }
    if (any_object) {
        out.vertex_begin.push_back((int64_t)out.vertices.size() / 3);
        out.tri_begin.push_back((int64_t)out.triangles.size() / 3);
    } else {                                             // nothing ever started: upstream returns []
        out.vertices.clear(); out.triangles.clear();
    }
    return true;
}

}  // namespace rt
