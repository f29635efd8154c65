// ref_harness.cpp - drives the UNMODIFIED reference renderer as the parity oracle.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit textually includes /root/reference/main.cpp
// (via -I, nothing is copied into this repository) with `main` renamed, then calls the reference's
// own InitParams / MakeCamera / ParseOBJ / CalculateTangents / BuildHierarchy / InitScene /
// MakeMaterial / RenderPixel.  It is BUILT only in the build container (/root/reference does not exist on the GPU box); the
// binary travels there as a checker, and since round 4 bench.py times it on the GPU box's host cores (cpu_baseline kind
// "reference", one pinned process per core: --rows k n) beside the port.  The product never depends on it.
//
// What it adds on top of the reference:
//   * fixed samples-per-pixel: RenderSharedData{min_samples = max_samples = 1} and one RenderPixel
//     call per (pixel, sample), summed in sample order and divided by spp - float-for-float what
//     RenderPixel does with min = max = spp (main.cpp:237-263);
//   * per-(pixel,sample) reseed Random_Seed(&job.rng, prt_sample_key(seed, pixel, sample)) so pixels
//     are independent of the rank count and of each other;
//   * an optional sparse pixel lattice, raw f32 output, scene / sphere-tree dumps and per-function
//     known-answer vectors for the CPU restatement's unit tests;
//   * --dump-desc: FlattenReferenceScene (include/prt_flatten_ref.h) run over the reference's own Scene graph, every
//     array of the resulting prt_scene_desc written out - what a maintainer's RenderTask would hand to prt_upload_scene;
//   * --write-png: the reference's own WriteFramebufferImage (LogAverageLuma + Color_Pack + stbi_write_png,
//     main.cpp:78-131) on the rendered frame, and LogAverageLuma's value in the stats - pins the output path.
//
// Build: see oracle/Makefile (flags follow the reference's build.sh:6).

#define main ref_main
#include "main.cpp"
#undef main

#include "../include/prt_key.h"
// the level-1 drop-in's flattening, compiled here against the reference's REAL scene types
#include "../include/prt_flatten_ref.h"

#include <string>
#include <time.h>

namespace {

double NowSeconds() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

u32 FloatBits(float f) {
    u32 u;
    memcpy(&u, &f, 4);
    return u;
}

struct Section {
    FILE * fp;
    void Put(const char * name, const void * data, size_t bytes) {
        u32 len = (u32)strlen(name);
        u64 n = bytes;
        fwrite(&len, 4, 1, fp);
        fwrite(name, 1, len, fp);
        fwrite(&n, 8, 1, fp);
        if (bytes) fwrite(data, 1, bytes, fp);
    }
    template <typename T> void PutVec(const char * name, const std::vector<T> & v) {
        Put(name, v.empty() ? NULL : &v[0], v.size() * sizeof(T));
    }
};

struct HarnessArgs {
    const char * obj_name = "sponza.obj";
    u32 spp = 1;
    u32 adaptive_max = 0;            // > spp: the reference's own adaptive loop (main.cpp:245-258), one RNG stream per pixel
    u64 seed = 1234;
    u32 lattice = 1;
    u32 light_mode = 0;
    const char * out = NULL;
    const char * dump_scene = NULL;
    const char * kat = NULL;
    const char * stats = NULL;
    const char * dump_desc = NULL;
    const char * write_png = NULL;
    u32 rows_k = 0, rows_n = 1;      // --rows k n: render only the lattice rows ly with ly % n == k (the others stay zero): n processes,
                                     // one per core, are how bench.py times the reference on all cores of the GPU box's host, as
                                     // `mpirun -n N` runs the reference's own ranks side by side (main.cpp:311-347)
};

HarnessArgs ParseHarnessArgs(int argc, char ** argv) {
    HarnessArgs a;
    for (int i = 1; i < argc; ++i) {
        std::string s = argv[i];
        bool has_val = i + 1 < argc;
        if (s == "--obj" && has_val) a.obj_name = argv[++i];
        else if (s == "--spp" && has_val) a.spp = (u32)atoi(argv[++i]);
        else if (s == "--adaptive" && has_val) a.adaptive_max = (u32)atoi(argv[++i]);
        else if (s == "--seed" && has_val) a.seed = strtoull(argv[++i], NULL, 0);
        else if (s == "--lattice" && has_val) a.lattice = (u32)atoi(argv[++i]);
        else if (s == "--light-mode" && has_val) a.light_mode = (u32)atoi(argv[++i]);
        else if (s == "--out" && has_val) a.out = argv[++i];
        else if (s == "--dump-scene" && has_val) a.dump_scene = argv[++i];
        else if (s == "--kat" && has_val) a.kat = argv[++i];
        else if (s == "--stats" && has_val) a.stats = argv[++i];
        else if (s == "--dump-desc" && has_val) a.dump_desc = argv[++i];
        else if (s == "--write-png" && has_val) a.write_png = argv[++i];
        else if (s == "--rows" && i + 2 < argc) { a.rows_k = (u32)atoi(argv[++i]); a.rows_n = (u32)atoi(argv[++i]); if (a.rows_n < 1) a.rows_n = 1; }
    }
    return a;
}

// ---------------------------------------------------------------------------------------------
// Known-answer vectors for single functions of the hot path.
// ---------------------------------------------------------------------------------------------

void WriteKnownAnswers(const char * path, Camera * cam) {
    FILE * fp = fopen(path, "wb");
    Section out = { fp };

    // A17: RandomState.
    {
        u64 seeds[] = { 0ULL, 0x835fdd9143716fe3ULL, 1ULL, 0xFFFFFFFFFFFFFFFFULL, 0x0123456789abcdefULL,
                        prt_sample_key(1234, 0, 0), prt_sample_key(1234, 777, 3) };
        std::vector<u64> seed_v, state_v, next_v;
        std::vector<float> f01_v, f11_v;
        for (u32 s = 0; s < array_count(seeds); ++s) {
            RandomState r;
            Random_Seed(&r, seeds[s]);
            seed_v.push_back(seeds[s]);
            for (u32 i = 0; i < 16; ++i) state_v.push_back(r._state[i]);
            for (u32 i = 0; i < 40; ++i) next_v.push_back(Random_Next(&r));
            Random_Seed(&r, seeds[s]);
            for (u32 i = 0; i < 24; ++i) f01_v.push_back(Random_NextFloat01(&r));
            Random_Seed(&r, seeds[s]);
            for (u32 i = 0; i < 24; ++i) f11_v.push_back(Random_NextFloat11(&r));
        }
        out.PutVec("rng_seed", seed_v);
        out.PutVec("rng_state", state_v);
        out.PutVec("rng_next", next_v);
        out.PutVec("rng_f01", f01_v);
        out.PutVec("rng_f11", f11_v);
        // the key function itself
        std::vector<u64> keys;
        for (u32 p = 0; p < 8; ++p) for (u32 s = 0; s < 4; ++s) keys.push_back(prt_sample_key(1234, p * 1000003u, s));
        out.PutVec("key_1234", keys);
    }

    RandomState r;
    Random_Seed(&r, 0xC0FFEEULL);
    auto rnd = [&](float lo, float hi) { return lo + (hi - lo) * Random_NextFloat01(&r); };

    // A8/A10: Hammersley + cosine-weighted bounce directions for a set of normals.
    {
        std::vector<float> xi, normals, dirs;
        for (u32 i = 0; i < 1024; ++i) {
            Vector2 h = Hammersley(i, 1024);
            xi.push_back(h.x); xi.push_back(h.y);
        }
        Vector3 ns[] = { Vector3(0, 0, 1), Vector3(0, 1, 0), Vector3(1, 0, 0), Vector3(0, 0, -1),
                         Normalize(Vector3(0.3f, 0.9f, -0.2f)), Normalize(Vector3(-0.01f, 0.004f, 0.99995f)),
                         Normalize(Vector3(-0.5f, -0.5f, 0.7f)) };
        for (u32 k = 0; k < array_count(ns); ++k) {
            normals.push_back(ns[k].x); normals.push_back(ns[k].y); normals.push_back(ns[k].z);
            for (u32 i = 0; i < 1024; ++i) {
                Ray ray = GetDiffuseReflectionRay(Vector3(1, 2, 3), ns[k], Hammersley(i, 1024));
                dirs.push_back(ray.direction.x); dirs.push_back(ray.direction.y); dirs.push_back(ray.direction.z);
            }
        }
        out.PutVec("hammersley_1024", xi);
        out.PutVec("diffuse_normals", normals);
        out.PutVec("diffuse_dirs", dirs);
    }

    // A9: Phong lobe directions: (normal, Ns, samp, spec_samples) -> direction.
    {
        std::vector<float> in, dirs;
        float ns_vals[] = { 10.0f, 6.0f, 40.0f, 0.0f };
        u32 counts[] = { 1, 2, 8 };
        for (u32 a = 0; a < array_count(ns_vals); ++a)
        for (u32 c = 0; c < array_count(counts); ++c)
        for (u32 samp = 0; samp < counts[c]; ++samp) {
            Vector3 n = Normalize(Vector3(rnd(-1, 1), rnd(-1, 1), rnd(-1, 1)));
            Ray ray = GetSpecularReflectionRay(Vector3(0, 0, 0), n, ns_vals[a], Hammersley(samp, counts[c]));
            in.push_back(n.x); in.push_back(n.y); in.push_back(n.z);
            in.push_back(ns_vals[a]); in.push_back((float)samp); in.push_back((float)counts[c]);
            dirs.push_back(ray.direction.x); dirs.push_back(ray.direction.y); dirs.push_back(ray.direction.z);
        }
        out.PutVec("spec_in", in);
        out.PutVec("spec_dirs", dirs);
    }

    // A11: Fresnel.
    {
        std::vector<float> in, res;
        float nis[] = { 1.5f, 1.45f, 1.8f, 1.0f, 0.0f, 0.7f };
        for (u32 a = 0; a < array_count(nis); ++a)
        for (u32 k = 0; k < 16; ++k) {
            Vector3 n = Normalize(Vector3(rnd(-1, 1), rnd(-1, 1), rnd(-1, 1)));
            Vector3 d = Normalize(Vector3(rnd(-1, 1), rnd(-1, 1), rnd(-1, 1)));
            in.push_back(nis[a]);
            in.push_back(n.x); in.push_back(n.y); in.push_back(n.z);
            in.push_back(d.x); in.push_back(d.y); in.push_back(d.z);
            res.push_back(FresnelAmount(1.0f, nis[a], n, d));
        }
        out.PutVec("fresnel_in", in);
        out.PutVec("fresnel_out", res);
    }

    // A4: ray / triangle.  Rays are aimed near the triangle so that about half of them hit.
    {
        std::vector<float> in, res;
        for (u32 k = 0; k < 4096; ++k) {
            Vector3 a(rnd(-4, 4), rnd(-4, 4), rnd(-4, 4));
            Vector3 b = a + Vector3(rnd(-2, 2), rnd(-2, 2), rnd(-2, 2));
            Vector3 c = a + Vector3(rnd(-2, 2), rnd(-2, 2), rnd(-2, 2));
            float u = rnd(-0.3f, 1.0f), v = rnd(-0.3f, 1.0f);
            Vector3 target = a + (b - a) * u + (c - a) * v;
            Ray ray;
            ray.origin = Vector3(rnd(-8, 8), rnd(-8, 8), rnd(-8, 8));
            ray.direction = Normalize(target - ray.origin);
            if (k % 7 == 0) ray.direction = ray.direction * -1.0f;
            float max_t = (k % 5 == 0) ? rnd(0.0f, 12.0f) : FLT_MAX;
            RaycastHit hit = { max_t };
            bool ok = IntersectRayTriangle(ray, a, b, c, &hit);
            float rec[] = { ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z,
                            a.x, a.y, a.z, b.x, b.y, b.z, c.x, c.y, c.z, max_t };
            in.insert(in.end(), rec, rec + array_count(rec));
            float o[] = { ok ? 1.0f : 0.0f, hit.t, hit.bw.x, hit.bw.y, hit.bw.z,
                          hit.position.x, hit.position.y, hit.position.z, hit.normal.x, hit.normal.y, hit.normal.z };
            if (!ok) { for (u32 j = 1; j < array_count(o); ++j) o[j] = 0.0f; }
            res.insert(res.end(), o, o + array_count(o));
        }
        out.PutVec("tri_in", in);
        out.PutVec("tri_out", res);
    }

    // A2: ray / sphere.
    {
        std::vector<float> in, res;
        for (u32 k = 0; k < 2048; ++k) {
            Sphere s;
            s.center = Vector3(rnd(-4, 4), rnd(-4, 4), rnd(-4, 4));
            s.radius = rnd(0.1f, 3.0f);
            Ray ray;
            ray.origin = Vector3(rnd(-6, 6), rnd(-6, 6), rnd(-6, 6));
            Vector3 target = s.center + Vector3(rnd(-1, 1), rnd(-1, 1), rnd(-1, 1)) * (s.radius * 1.3f);
            ray.direction = Normalize(target - ray.origin);
            if (k % 5 == 0) ray.direction = ray.direction * -1.0f;
            RaycastHit hit = {};
            bool ok = IntersectRaySphere(ray, s, &hit);
            float rec[] = { ray.origin.x, ray.origin.y, ray.origin.z, ray.direction.x, ray.direction.y, ray.direction.z,
                            s.center.x, s.center.y, s.center.z, s.radius };
            in.insert(in.end(), rec, rec + array_count(rec));
            res.push_back(ok ? 1.0f : 0.0f);
            res.push_back(ok ? hit.t : 0.0f);
        }
        out.PutVec("sphere_in", in);
        out.PutVec("sphere_out", res);
    }

    // A13: camera rays for the camera the command line configured.
    {
        std::vector<float> in, res;
        float camrec[] = { cam->tan_a2, cam->aspect, cam->inv_width, cam->inv_height,
                           cam->camera_position.x, cam->camera_position.y, cam->camera_position.z,
                           cam->camera_forward.x, cam->camera_forward.y, cam->camera_forward.z,
                           cam->camera_right.x, cam->camera_right.y, cam->camera_right.z,
                           cam->camera_up.x, cam->camera_up.y, cam->camera_up.z };
        out.Put("camera", camrec, sizeof(camrec));
        for (u32 k = 0; k < 256; ++k) {
            Vector2 p(rnd(-1.0f, (float)gParams.image_width + 1.0f), rnd(-1.0f, (float)gParams.image_height + 1.0f));
            Ray ray = MakeCameraRay(cam, p);
            in.push_back(p.x); in.push_back(p.y);
            res.push_back(ray.direction.x); res.push_back(ray.direction.y); res.push_back(ray.direction.z);
        }
        out.PutVec("camray_in", in);
        out.PutVec("camray_out", res);
    }
    fclose(fp);
}

// ---------------------------------------------------------------------------------------------
// Scene + sphere tree dump (validates the product's OBJ loader and hierarchy builder bit for bit).
// ---------------------------------------------------------------------------------------------

void DumpScene(const char * path, Mesh * mesh, BoundingHierarchy * h, Scene * scene) {
    FILE * fp = fopen(path, "wb");
    Section out = { fp };
    out.Put("positions", mesh->positions.empty() ? NULL : &mesh->positions[0], mesh->positions.size() * sizeof(Vector3));
    out.Put("texcoords", mesh->texcoords.empty() ? NULL : &mesh->texcoords[0], mesh->texcoords.size() * sizeof(Vector2));
    out.Put("normals", mesh->normals.empty() ? NULL : &mesh->normals[0], mesh->normals.size() * sizeof(Vector3));
    std::vector<u32> group_sizes, idx_p, idx_t, idx_n;
    std::vector<float> group_mats;
    std::string names;
    for (u32 g = 0; g < mesh->groups.size(); ++g) {
        MeshGroup * mg = &mesh->groups[g];
        group_sizes.push_back((u32)mg->idx_positions.size());
        idx_p.insert(idx_p.end(), mg->idx_positions.begin(), mg->idx_positions.end());
        idx_t.insert(idx_t.end(), mg->idx_texcoords.begin(), mg->idx_texcoords.end());
        idx_n.insert(idx_n.end(), mg->idx_normals.begin(), mg->idx_normals.end());
        Material * m = mg->material ? mg->material : scene->default_mat;
        float rec[] = { m->specular_intensity, m->index_of_refraction, m->alpha,
                        m->ambient_color.x, m->ambient_color.y, m->ambient_color.z, m->ambient_color.w,
                        m->diffuse_color.x, m->diffuse_color.y, m->diffuse_color.z, m->diffuse_color.w,
                        m->specular_color.x, m->specular_color.y, m->specular_color.z, m->specular_color.w,
                        mg->material ? 1.0f : 0.0f };
        group_mats.insert(group_mats.end(), rec, rec + array_count(rec));
        names += mg->name ? mg->name : "";
        names += "\n";
    }
    // texture slots of every group's material as the reference decoded them: (size_x, size_y, channels) + texel bytes,
    // slot order ambient, diffuse, specular, alpha, bump (the bump slot holds the CONVERTED normal map)
    {
        std::vector<u32> tex_dims;
        std::vector<u8> tex_bytes;
        for (u32 g = 0; g < mesh->groups.size(); ++g) {
            Material * m = mesh->groups[g].material ? mesh->groups[g].material : scene->default_mat;
            Texture * slots[5] = { m->ambient_texture, m->diffuse_texture, m->specular_texture, m->alpha_texture, m->bump_texture };
            for (u32 k = 0; k < 5; ++k) {
                Texture * t = slots[k];
                tex_dims.push_back(t ? t->size_x : 0);
                tex_dims.push_back(t ? t->size_y : 0);
                tex_dims.push_back(t ? t->channels : 0);
                if (t) tex_bytes.insert(tex_bytes.end(), t->texels, t->texels + (size_t)t->size_x * t->size_y * t->channels);
            }
        }
        out.PutVec("group_texture_dims", tex_dims);
        out.PutVec("group_texture_bytes", tex_bytes);
    }
    out.Put("tangents", mesh->tangents.empty() ? NULL : &mesh->tangents[0], mesh->tangents.size() * sizeof(Vector3));
    out.PutVec("group_index_counts", group_sizes);
    out.PutVec("idx_positions", idx_p);
    out.PutVec("idx_texcoords", idx_t);
    out.PutVec("idx_normals", idx_n);
    out.PutVec("group_materials", group_mats);
    out.Put("group_names", names.data(), names.size());

    std::vector<float> spheres;
    std::vector<u32> children;
    std::vector<s32> node_group;
    for (u32 i = 0; i < h->spheres.size(); ++i) {
        BoundingSphere s = h->spheres[i];
        spheres.push_back(s.s.center.x); spheres.push_back(s.s.center.y); spheres.push_back(s.s.center.z);
        spheres.push_back(s.s.radius);
        children.push_back(s.c0); children.push_back(s.c1);
        MeshGroup * mg = h->mesh_groups[i];
        node_group.push_back(mg ? (s32)(mg - &mesh->groups[0]) : -1);
    }
    out.PutVec("spheres", spheres);
    out.PutVec("sphere_children", children);
    out.PutVec("sphere_group", node_group);
    fclose(fp);
}

// Every array of the prt_scene_desc FlattenReferenceScene builds from the reference's scene graph.
void DumpDesc(const char * path, Scene * scene) {
    RefFlatScene flat;
    FlattenReferenceScene(scene, &flat);
    FILE * fp = fopen(path, "wb");
    Section out = { fp };
    out.PutVec("positions", flat.positions);
    out.PutVec("normals", flat.normals);
    out.PutVec("texcoords", flat.texcoords);
    out.PutVec("tangents", flat.tangents);
    out.PutVec("idx_positions", flat.idx_positions);
    out.PutVec("idx_texcoords", flat.idx_texcoords);
    out.PutVec("idx_normals", flat.idx_normals);
    out.PutVec("groups", flat.groups);               // prt_group records, 12 bytes
    out.PutVec("materials", flat.materials);         // prt_material records, 80 bytes
    out.PutVec("lights", flat.lights);               // prt_light records, 48 bytes
    out.PutVec("spheres", flat.spheres);             // prt_bsphere records, 24 bytes
    out.PutVec("sphere_group", flat.sphere_group);
    std::vector<u32> dims;
    std::vector<u8> texels;
    for (size_t i = 0; i < flat.textures.size(); ++i) {
        const prt_texture & t = flat.textures[i];
        dims.push_back(t.size_x); dims.push_back(t.size_y); dims.push_back(t.channels);
        texels.insert(texels.end(), t.texels, t.texels + (size_t)t.size_x * t.size_y * t.channels);
    }
    out.PutVec("texture_dims", dims);
    out.PutVec("texture_bytes", texels);
    fclose(fp);
}

}  // namespace

int main(int argc, char ** argv) {
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &gMPI_CommSize);
    MPI_Comm_rank(MPI_COMM_WORLD, &gMPI_CommRank);

    HarnessArgs args = ParseHarnessArgs(argc, argv);
    InitParams(argc, argv);                                   // reference flag parser (main.cpp:416-504)

    // Same call sequence as the reference's main (main.cpp:546-599).
    Camera cam = MakeCamera(gParams.camera_fov, gParams.image_width, gParams.image_height);
    Matrix33 transform;
    transform.SetIdentity();
    Mesh * mesh = ParseOBJ(gParams.data_dirname, (char *)args.obj_name, transform);
    if (!mesh) {
        fprintf(stderr, "ref_harness: cannot load %s from %s\n", args.obj_name, gParams.data_dirname);
        return 2;
    }
    CalculateTangents(mesh);
    BoundingHierarchy hierarchy;
    double t_build0 = NowSeconds();
    BuildHierarchy(&hierarchy, mesh);
    double t_build = NowSeconds() - t_build0;

    u32 total_tris = 0;
    Scene scene = InitScene();
    if (args.light_mode == 1) {
        scene.light_count = 2;                                // second directional light (main.cpp:526-528)
    } else if (args.light_mode == 2) {
        scene.light_count = 2;                                // directional + point (exercises raytracer.cpp:391-405)
        scene.lights[1].type = Light_Point;
        scene.lights[1].color = Vector4(1.0f, 0.85f, 0.6f, 1.0f) * 6.0f;
        scene.lights[1].position = gParams.camera_position + Vector3(0.5f, 1.0f, -2.0f);
        scene.lights[1].falloff = 3.0f;
    }
    scene.hierarchy = &hierarchy;
    scene.default_mat = MakeMaterial(Vector4(0.75f, 0.5f, 0.75f, 1.0f));
    for (u32 i = 0; i < hierarchy.mesh_groups.size(); ++i) {
        MeshGroup * mg = hierarchy.mesh_groups[i];
        SceneObject * obj = (SceneObject *)calloc(1, sizeof(SceneObject));
        obj->mesh_group = mg;
        obj->mesh = mesh;
        obj->type = ObjectType_MeshGroup;
        obj->material = scene.default_mat;
        if (mg) {
            total_tris += mg->idx_positions.size() / 3;
            if (mg->material) obj->material = mg->material;
        }
        scene.objects.push_back(obj);
    }

    if (args.dump_scene) DumpScene(args.dump_scene, mesh, &hierarchy, &scene);
    if (args.dump_desc) DumpDesc(args.dump_desc, &scene);
    if (args.kat) WriteKnownAnswers(args.kat, &cam);

    u32 w = gParams.image_width, h = gParams.image_height;
    u32 lw = (w + args.lattice - 1) / args.lattice, lh = (h + args.lattice - 1) / args.lattice;
    DebugCounters debug = {};
    double render_s = 0.0;
    float scene_luma = 0.0f;
    if (args.out) {
        RenderSharedData shared;
        shared.cam = &cam;
        shared.scene = &scene;
        shared.width = w;
        shared.height = h;
        shared.min_samples = 1;
        shared.max_samples = 1;
        RenderJob job;
        job.shared = &shared;
        job.start_idx = 0;
        job.end_idx = w * h;
        job.buffer = NULL;

        std::vector<Vector4> pixels((size_t)lw * lh);
        double t0 = NowSeconds();
        for (u32 ly = 0; ly < lh; ++ly) {
            if (ly % args.rows_n != args.rows_k) continue;
            for (u32 lx = 0; lx < lw; ++lx) {
                u32 x = lx * args.lattice, y = ly * args.lattice;
                u32 pixel = y * w + x;
                if (args.adaptive_max > args.spp) {
                    // adaptive mode: RenderPixel exactly as main.cpp:224-265 runs it (min_samples fixed samples, then
                    // up to max_samples with the variance rule), on one RNG stream seeded per PIXEL
                    shared.min_samples = args.spp;
                    shared.max_samples = args.adaptive_max;
                    Random_Seed(&job.rng, prt_sample_key(args.seed, pixel, 0));
                    pixels[(size_t)ly * lw + lx] = RenderPixel(&job, &debug, x, y);
                    continue;
                }
                Vector4 sum;
                for (u32 s = 0; s < args.spp; ++s) {
                    Random_Seed(&job.rng, prt_sample_key(args.seed, pixel, s));
                    sum += RenderPixel(&job, &debug, x, y);
                }
                sum /= args.spp;
                sum.w = 1.0f;
                pixels[(size_t)ly * lw + lx] = sum;
            }
        }
        render_s = NowSeconds() - t0;
        FILE * fp = fopen(args.out, "wb");
        fwrite(&pixels[0], sizeof(Vector4), pixels.size(), fp);
        fclose(fp);
        if (args.write_png && args.lattice == 1) {
            // the reference's output path on this frame, untouched: main.cpp:101-131 (and :78-99, color.h:105-111)
            Framebuffer fb;
            fb.pixels = &pixels[0];
            fb.width = w;
            fb.height = h;
            scene_luma = LogAverageLuma(&fb);
            WriteFramebufferImage(&fb, (char *)args.write_png);
        }
    }

    FILE * sf = args.stats ? fopen(args.stats, "w") : stdout;
    fprintf(sf, "{\"triangles\": %u, \"groups\": %u, \"spheres\": %u, \"width\": %u, \"height\": %u, \"lattice\": %u, "
                "\"lattice_width\": %u, \"lattice_height\": %u, \"spp\": %u, \"seed\": %llu, \"bounce_depth\": %u, "
                "\"ray_count\": %llu, \"sphere_check_count\": %llu, \"mesh_check_count\": %llu, "
                "\"render_seconds\": %.6f, \"hierarchy_seconds\": %.6f, \"scene_luma_bits\": %u}\n",
            total_tris, (u32)mesh->groups.size(), (u32)hierarchy.spheres.size(), w, h, args.lattice, lw, lh, args.spp,
            (unsigned long long)args.seed, gParams.bounce_depth,
            (unsigned long long)debug.ray_count, (unsigned long long)debug.sphere_check_count,
            (unsigned long long)debug.mesh_check_count, render_s, t_build, FloatBits(scene_luma));
    if (args.stats) fclose(sf);

    MPI_Finalize();
    return 0;
}
