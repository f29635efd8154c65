// obj_loader.cpp - OBJ / MTL loader behind the reference's ParseOBJ signature.
//
// New implementation; what it preserves is the observable behaviour of obj_parser.cpp
// so that the same file yields the same Mesh, index for index and bit for bit:
//   * record dispatch on the first non-blank character of each line; "v" must be followed by a
//     space (obj_parser.cpp:367-384); numbers via strtof / strtol;
//   * faces are p/t/n triples (obj_parser.cpp:130-140), <= 8 corners, fan-triangulated around corner 0
//     (:169-194); indices <= 0 are relative to the current array size (:142-159);
//   * `usemtl` on a group that already has a material starts a new group with the same name (:399-404);
//     an unknown material name maps to NULL (the scene's default material is used, main.cpp:586-589);
//   * MTL materials start zeroed (:255) - a missing `d` gives alpha 0; any line starting with `d` sets
//     alpha (:273-275); Ka/Kd/Ks/Ke get w = 1 (ReadColor, :108-112).
// Differences, all in behaviour the reference leaves undefined: no chdir (paths are joined instead),
// and malformed input (face before any `g`, statement before `newmtl`) is reported and skipped instead
// of dereferencing NULL (:387-390).  Texture maps (`map_*`) are decoded by image_in.cpp.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "prt_scene.h"

namespace {

bool StartsWith(const char * s, const char * prefix) { return strncmp(prefix, s, strlen(prefix)) == 0; }

// Skip blanks but never a newline (obj_parser.cpp:48-54).
const char * SkipBlanks(const char * p) {
    while (*p && *p != '\n' && isspace((unsigned char)*p)) ++p;
    return p;
}

const char * SkipLine(const char * p, u32 * line) {
    while (*p && *p != '\n') ++p;
    if (*p == '\n') { ++p; ++*line; }
    return p;
}

struct NumberReader {
    const char * p;
    u32 line;
    float Float() {
        char * end;
        p = SkipBlanks(p);
        float v = strtof(p, &end);
        if (!isspace((unsigned char)*end))
            fprintf(stderr, "obj: expected whitespace after number on line %u\n", line);
        p = end;
        return v;
    }
    Vector3 Vec3() { Vector3 v; v.x = Float(); v.y = Float(); v.z = Float(); return v; }
    Vector2 Vec2() { Vector2 v; v.x = Float(); v.y = Float(); return v; }
};

char * CopyToken(const char * p) {
    const char * b = SkipBlanks(p);
    const char * e = b;
    while (*e && !isspace((unsigned char)*e)) ++e;
    if (e == b) return NULL;
    char * out = (char *)malloc((size_t)(e - b) + 1);
    memcpy(out, b, (size_t)(e - b));
    out[e - b] = '\0';
    return out;
}

char * SlurpFile(const std::string & path) {
    FILE * fp = fopen(path.c_str(), "rb");
    if (!fp) return NULL;
    // a directory, FIFO or other unseekable path makes fseek / ftell fail (ftell = -1): not a file we can read
    long n = -1;
    if (fseek(fp, 0, SEEK_END) == 0) n = ftell(fp);
    if (n < 0 || (unsigned long)n > (1ul << 40) || fseek(fp, 0, SEEK_SET) != 0) { fclose(fp); return NULL; }
    char * bytes = (char *)calloc(1, (size_t)n + 1);
    if (!bytes) { fclose(fp); return NULL; }
    size_t got = fread(bytes, 1, (size_t)n, fp);
    bytes[got <= (size_t)n ? got : (size_t)n] = '\0';
    fclose(fp);
    return bytes;
}

std::string JoinPath(const char * dir, const char * name) {
    if (!dir || !*dir || name[0] == '/') return name;
    std::string out = dir;
    if (out[out.size() - 1] != '/') out += '/';
    return out + name;
}

u32 ResolveIndex(long idx, size_t count) {
    return idx > 0 ? (u32)(idx - 1) : (u32)((s32)count + (s32)idx);
}

void ReadFaceRecord(const char * p, u32 line, Mesh * mesh, MeshGroup * group) {
    u32 pos[8], tex[8], nrm[8];
    u32 corners = 0;
    p = SkipBlanks(p);
    while (*p && *p != '\n' && *p != '\r') {
        if (corners >= 8) {
            fprintf(stderr, "obj: more than 8 corners on line %u, rest ignored\n", line);
            break;
        }
        // p/t/n (obj_parser.cpp:130-140).  The reference steps over whatever separates the numbers; here a corner that is
        // not three '/'-separated integers ends the face (never read past the end of the buffer, always move forward).
        char * end;
        long ip = strtol(p, &end, 10);
        bool ok = end != p && *end == '/';
        long it = 0, in = 0;
        if (ok) {
            const char * q = end + 1;
            it = strtol(q, &end, 10);
            ok = end != q && *end == '/';
        }
        if (ok) {
            const char * q = end + 1;
            in = strtol(q, &end, 10);
            ok = end != q;
        }
        if (!ok) {
            fprintf(stderr, "obj: malformed face corner on line %u (expected p/t/n), rest of the face ignored\n", line);
            break;
        }
        if (*end && !isspace((unsigned char)*end)) fprintf(stderr, "obj: expected whitespace after face corner on line %u\n", line);
        pos[corners] = ResolveIndex(ip, mesh->positions.size());
        tex[corners] = ResolveIndex(it, mesh->texcoords.size());
        nrm[corners] = ResolveIndex(in, mesh->normals.size());
        ++corners;
        p = SkipBlanks(end);
    }
    for (u32 i = 1; i + 1 < corners; ++i) {
        const u32 k[3] = { 0, i, i + 1 };
        for (u32 c = 0; c < 3; ++c) {
            group->idx_positions.push_back(pos[k[c]]);
            group->idx_texcoords.push_back(tex[k[c]]);
            group->idx_normals.push_back(nrm[k[c]]);
        }
    }
}

Vector4 ReadColourRecord(const char * p, u32 line) {
    NumberReader r = { p, line };
    Vector3 rgb = r.Vec3();
    return Vector4(rgb.x, rgb.y, rgb.z, 1.0f);
}

// map_* records (obj_parser.cpp:303-331).  The file name is the first token after the keyword; paths are
// relative to the OBJ's directory (the reference chdir()s there, obj_parser.cpp:353).  A bump map is a height map
// that is converted to a normal map on load (:322-329); an image the decoder rejects leaves the slot empty.
Texture * LoadMap(const char * working_dir, const char * after_keyword, u32 line) {
    char * name = CopyToken(after_keyword);
    if (!name) {
        fprintf(stderr, "mtl: texture map without a file name on line %u\n", line);
        return NULL;
    }
    Texture * t = LoadTexture(JoinPath(working_dir, name).c_str());
    free(name);
    return t;
}

MaterialLibrary * ParseMTL(const char * working_dir, const std::string & path) {
    char * bytes = SlurpFile(path);
    MaterialLibrary * lib = new MaterialLibrary;
    if (!bytes) {
        fprintf(stderr, "obj: cannot open material library %s\n", path.c_str());
        return lib;
    }
    Material * mat = NULL;
    u32 line = 1;
    for (const char * p = bytes; *p; p = SkipLine(p, &line)) {
        p = SkipBlanks(p);
        char c = *p;
        if (c == 'n') {
            if (StartsWith(p, "newmtl")) {
                char * name = CopyToken(p + 6);
                mat = (Material *)calloc(1, sizeof(Material));
                mat->name = name;
                if (name) lib->materials[name] = mat;
                lib->in_file_order.push_back(mat);
            }
            continue;
        }
        if (c != 'N' && c != 'd' && c != 'K' && c != 'm') continue;     // 'T', 'i' and others: ignored
        if (!mat) {
            fprintf(stderr, "mtl: statement before any newmtl on line %u, skipped\n", line);
            continue;
        }
        NumberReader r = { p + 2, line };
        if (c == 'N') {
            if (p[1] == 's') mat->specular_intensity = r.Float();
            else if (p[1] == 'i') mat->index_of_refraction = r.Float();
        } else if (c == 'd') {
            r.p = p + 1;
            mat->alpha = r.Float();
        } else if (c == 'K') {
            if (p[1] == 'a') mat->ambient_color = ReadColourRecord(p + 2, line);
            else if (p[1] == 'd') mat->diffuse_color = ReadColourRecord(p + 2, line);
            else if (p[1] == 's') mat->specular_color = ReadColourRecord(p + 2, line);
            else if (p[1] == 'e') mat->emissive_color = ReadColourRecord(p + 2, line);
        } else if (c == 'm' && StartsWith(p, "map_")) {
            const char * q = p + 4;
            if (StartsWith(q, "Ka")) mat->ambient_texture = LoadMap(working_dir, q + 2, line);
            else if (StartsWith(q, "Kd")) mat->diffuse_texture = LoadMap(working_dir, q + 2, line);
            else if (StartsWith(q, "Ks")) mat->specular_texture = LoadMap(working_dir, q + 2, line);
            else if (StartsWith(q, "d")) mat->alpha_texture = LoadMap(working_dir, q + 1, line);
            else if (StartsWith(q, "bump")) {
                Texture * height = LoadMap(working_dir, q + 4, line);
                if (height) {                                   // (the reference dereferences NULL here)
                    mat->bump_texture = ConvertHeightMapToNormalMap(height);
                    FreeTexture(height);
                }
            }
        }
    }
    free(bytes);
    return lib;
}

MeshGroup * AppendGroup(Mesh * mesh, char * name) {
    mesh->groups.push_back(MeshGroup());
    MeshGroup * g = &mesh->groups.back();
    g->name = name;
    return g;
}

}  // namespace

Mesh * ParseOBJ(const char * working_dir, const char * filename, Matrix33 transform) {
    char * bytes = SlurpFile(JoinPath(working_dir, filename));
    if (!bytes) return NULL;

    Mesh * mesh = new Mesh;
    s64 current = -1;          // index, not pointer: groups is a growing vector
    u32 line = 1;
    for (const char * p = bytes; *p; p = SkipLine(p, &line)) {
        p = SkipBlanks(p);
        switch (*p) {
        case 'v': {
            NumberReader r = { p + 2, line };
            if (p[1] == ' ') mesh->positions.push_back(transform * r.Vec3());
            else if (p[1] == 't') mesh->texcoords.push_back(r.Vec2());
            else if (p[1] == 'n') mesh->normals.push_back(transform * r.Vec3());
        } break;
        case 'f':
            if (current < 0) {
                fprintf(stderr, "obj: face declared with no active group, line %u, skipped\n", line);
                break;
            }
            ReadFaceRecord(p + 1, line, mesh, &mesh->groups[(size_t)current]);
            break;
        case 'g':
            AppendGroup(mesh, CopyToken(p + 1));
            current = (s64)mesh->groups.size() - 1;
            break;
        case 'u':
        case 'm':
            if (StartsWith(p, "usemtl")) {
                if (current < 0 || !mesh->material_library) {
                    fprintf(stderr, "obj: usemtl without group or mtllib on line %u, skipped\n", line);
                    break;
                }
                if (mesh->groups[(size_t)current].material) {
                    const char * old_name = mesh->groups[(size_t)current].name;
                    AppendGroup(mesh, old_name ? strdup(old_name) : NULL);
                    current = (s64)mesh->groups.size() - 1;
                }
                char * name = CopyToken(p + 6);
                Material * m = NULL;
                if (name) {
                    std::unordered_map<std::string, Material *>::iterator it = mesh->material_library->materials.find(name);
                    if (it != mesh->material_library->materials.end()) m = it->second;
                    free(name);
                }
                mesh->groups[(size_t)current].material = m;
            } else if (StartsWith(p, "mtllib")) {
                char * name = CopyToken(p + 6);
                if (name && !mesh->material_library) mesh->material_library = ParseMTL(working_dir, JoinPath(working_dir, name));
                free(name);
            }
            break;
        default:
            break;
        }
    }
    free(bytes);
    // Indices may name vertices that appear later in the file, so they can only be checked now.  The reference never
    // checks them (out-of-range indices are wild reads in CalculateTangents, BuildHierarchy and the triangle test);
    // here such a file does not load.
    for (size_t g = 0; g < mesh->groups.size(); ++g) {
        const MeshGroup & mg = mesh->groups[g];
        for (size_t i = 0; i < mg.idx_positions.size(); ++i) {
            if (mg.idx_positions[i] >= mesh->positions.size() || mg.idx_texcoords[i] >= mesh->texcoords.size() ||
                mg.idx_normals[i] >= mesh->normals.size()) {
                fprintf(stderr, "obj: face index out of range in group %s (%zu positions, %zu texcoords, %zu normals)\n",
                        mg.name ? mg.name : "?", mesh->positions.size(), mesh->texcoords.size(), mesh->normals.size());
                delete mesh;
                return NULL;
            }
        }
    }
    return mesh;
}

// mesh.h:59-130.  Tangents only feed bump mapping (raytracer.cpp:468-502); without a bump texture on any
// material the reference's loop body never runs and every tangent stays the zero vector.
void CalculateTangents(Mesh * mesh) {
    mesh->tangents.assign(mesh->normals.size(), Vector3());
    for (size_t g = 0; g < mesh->groups.size(); ++g) {
        MeshGroup * mg = &mesh->groups[g];
        if (!mg->material || !mg->material->bump_texture) continue;
        for (size_t i = 0; i + 2 < mg->idx_positions.size(); i += 3) {
            Vector3 p0 = mesh->positions[mg->idx_positions[i]];
            Vector3 d1 = mesh->positions[mg->idx_positions[i + 1]] - p0;
            Vector3 d2 = mesh->positions[mg->idx_positions[i + 2]] - p0;
            Vector2 t0 = mesh->texcoords[mg->idx_texcoords[i]];
            Vector2 u1 = mesh->texcoords[mg->idx_texcoords[i + 1]] - t0;
            Vector2 u2 = mesh->texcoords[mg->idx_texcoords[i + 2]] - t0;
            float f = (u1.x * u2.y - u2.x * u1.y);
            if (f <= 1e-7) continue;
            f = 1.0f / f;
            Vector3 t(f * (u2.y * d1.x - u1.y * d2.x), f * (u2.y * d1.y - u1.y * d2.y), f * (u2.y * d1.z - u1.y * d2.z));
            for (u32 c = 0; c < 3; ++c) mesh->tangents[mg->idx_normals[i + c]] += t;
        }
    }
    for (size_t i = 0; i < mesh->tangents.size(); ++i) mesh->tangents[i] = Normalize(mesh->tangents[i]);
}
