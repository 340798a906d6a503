// raytracer.cpp -- command-line driver with the reference executable's flags, on top of the C-ABI.
//
// Stands where the reference's main() stands (src/main.cpp:349-764): same options (src/main.cpp:352-373),
// same console text (src/main.cpp:419-483, 213-214, 265, 563, 747-753), same resolve and 0xABGR packing
// (src/main.cpp:708-743, done on the GPU by rt_resolve), same kind of output file (8-bit RGBA PNG, rows
// top to bottom).  What differs, on purpose:
//   * the samples are rendered by rt_render on a HIP device; -t/--threads is accepted, printed and unused;
//   * the command-line parser is ours.  It follows the rules of the parser the reference vendors
//     (ext/cxxopts.hpp:1167-1290: `--name value`, `--name=value`, grouped short flags, an option with an
//     implicit value takes the next word unless that word starts with '-');
//   * the PNG is written by the small encoder below (stored deflate blocks), so the FILE bytes differ from
//     stb_image_write's while the decoded pixels are what the reference's resolve gives for the same sums;
//   * -p writes the normal / depth / uv passes the reference describes in its disabled pre-pass
//     (src/main.cpp:270-326) from rt_camera_rays + rt_trace_rays at pixel centres; the reference's own -p
//     dumps buffers it never fills;
//   * extra options, absent from the reference: --scene NAME, --seed N, --device N, --path auto|exact|fast, and
//     --devices LIST [--reduce-always] [--repeat K] (the samples shared by several GPUs of the node, one RCCL sum-reduce;
//     the multi-GPU context is built once and K images are rendered on it), and
//     --accum FILE [--save-every N]: progressive render with save and resume, which the reference lists as not done
//     (TODO.txt:5-17).  The file holds the float sums and the number of samples in them; -s is the TOTAL wanted.
// No CPU rendering path exists here: without a HIP device the program stops with the library's error.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "rt_amd.h"
#include "rt_host.h"

namespace {

// ---------------------------------------------------------------- options (src/main.cpp:352-373)
struct Opt {
    const char *shrt, *lng, *help;
    bool is_string;
    const char *def, *implicit; // nullptr = none
    std::string value;
    int count;
};

Opt g_opts[] = {
    {"w", "width", "Output image width", false, "1920", nullptr, "", 0},
    {"h", "height", "Output image height", false, "1080", nullptr, "", 0},
    {"s", "samples", "Number of samples per pixel", false, "256", nullptr, "", 0},
    {"S", "sub-samples", "Number of sub-samples per thread", false, "1", nullptr, "", 0},
    {"r", "recurse", "Number of bounces", false, "1", "50", "", 0},
    {"t", "threads", "Number of threads (unused: the GPU renders)", false, "1", nullptr, "", 0},
    {"", "tw", "Tile width", false, "1", nullptr, "", 0},
    {"", "th", "Tile height", false, "1", nullptr, "", 0},
    {"", "windowed", "Show a preview window", false, "0", "1", "", 0},
    {"i", "input", "Input filename", true, nullptr, nullptr, "", 0},
    {"o", "output", "Output filename", true, "out.png", nullptr, "", 0},
    {"p", "passes", "Output passes as separate files", false, "0", "1", "", 0},
    {"m", "multi", "Output one file for each sample", false, "0", "1", "", 0},
    {"", "rx", "ROI start x (from left)", false, "0", nullptr, "", 0},
    {"", "ry", "ROI start y (from top)", false, "0", nullptr, "", 0},
    {"", "rw", "ROI width", false, "1", nullptr, "", 0},
    {"", "rh", "ROI height", false, "1", nullptr, "", 0},
    {"x", "exit", "Exit without rendering", false, "0", "1", "", 0},
    {"v", "verbose", "Prints text", false, "0", "1", "", 0},
    {"V", "extra-verbose", "Prints extra text", false, "0", "1", "", 0},
    // not in the reference
    {"", "scene", "Built-in scene name", true, "cornell_box", nullptr, "", 0},
    {"", "seed", "Seed of the per-path random numbers", false, "1", nullptr, "", 0},
    {"", "device", "HIP device ordinal", false, "0", nullptr, "", 0},
    {"", "path", "auto | exact | fast", true, "auto", nullptr, "", 0},
    {"", "accum", "Accumulator file: resume from it if it exists, save to it", true, "", nullptr, "", 0},
    {"", "save-every", "With --accum: also save after every N samples", false, "0", nullptr, "", 0},
    {"", "devices", "GPUs that share the samples, e.g. 0-7 or 0,2,5 (one RCCL sum-reduce onto the first)", true, "", nullptr, "", 0},
    {"", "reduce-always", "With --devices: run the RCCL reduce on one device too", false, "0", "1", "", 0},
    {"", "repeat", "With --devices: render the image K times on one context and print each image's time", false, "1", nullptr, "", 0},
};
const int g_nopts = sizeof(g_opts) / sizeof(g_opts[0]);

[[noreturn]] void die(const std::string &msg)
{
    fprintf(stderr, "raytracer: %s\n", msg.c_str());
    exit(1);
}

Opt *find_opt(const std::string &name)
{
    for (int i = 0; i < g_nopts; ++i)
        if (name == g_opts[i].shrt || name == g_opts[i].lng) return &g_opts[i];
    return nullptr;
}

void store(Opt *o, const std::string &name, const std::string &text)
{
    if (!o->is_string) { // integer options: the whole word must be a number
        char *end = nullptr;
        strtol(text.c_str(), &end, 10);
        if (text.empty() || *end) die("Argument '" + text + "' failed to parse for option '" + name + "'");
    }
    o->value = text;
    ++o->count;
}

// the next-word rule of ext/cxxopts.hpp:1070-1102
void take_arg(int argc, char **argv, int &cur, Opt *o, const std::string &name)
{
    if (cur + 1 >= argc) {
        if (!o->implicit) die("Option '" + name + "' is missing an argument");
        store(o, name, o->implicit);
    } else if (argv[cur + 1][0] == '-' && o->implicit) {
        store(o, name, o->implicit);
    } else {
        store(o, name, argv[cur + 1]);
        ++cur;
    }
}

bool is_alnum(char c) { return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); }

void parse_args(int argc, char **argv)
{
    for (int cur = 1; cur < argc; ++cur) {
        const char *a = argv[cur];
        if (!strcmp(a, "--")) break;
        size_t len = strlen(a);
        if (len >= 4 && a[0] == '-' && a[1] == '-' && is_alnum(a[2])) { // --name or --name=value (name of 2+ characters)
            const char *eq = strchr(a, '=');
            std::string name = eq ? std::string(a + 2, eq) : std::string(a + 2);
            bool ok = name.size() >= 2;
            for (char c : name) ok = ok && (is_alnum(c) || c == '-' || c == '_');
            if (!ok) continue; // not an option word: ignored, as the reference ignores stray words
            Opt *o = find_opt(name);
            if (!o) die("Option '" + name + "' does not exist");
            if (eq)
                store(o, name, eq + 1);
            else
                take_arg(argc, argv, cur, o, name);
        } else if (len >= 2 && a[0] == '-' && a[1] != '-') { // -abc
            bool ok = true;
            for (size_t i = 1; i < len; ++i) ok = ok && is_alnum(a[i]);
            if (!ok) continue;
            for (size_t i = 1; i < len; ++i) {
                std::string name(1, a[i]);
                Opt *o = find_opt(name);
                if (!o || !*o->shrt) die("Option '" + name + "' does not exist");
                if (i + 1 == len)
                    take_arg(argc, argv, cur, o, name);
                else if (o->implicit)
                    store(o, name, o->implicit);
                else
                    die("Option '" + name + "' requires an argument");
            }
        }
        // anything else is a stray word and is ignored
    }
    for (int i = 0; i < g_nopts; ++i)
        if (!g_opts[i].count && g_opts[i].def) g_opts[i].value = g_opts[i].def;
}

int opt_int(const char *name) { return (int)strtol(find_opt(name)->value.c_str(), nullptr, 10); }
const std::string &opt_str(const char *name) { return find_opt(name)->value; }

// ---------------------------------------------------------------- PNG (8-bit RGBA, filter 0, stored deflate)
uint32_t g_crc_table[256];
void crc_init()
{
    for (uint32_t n = 0; n < 256; ++n) {
        uint32_t c = n;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        g_crc_table[n] = c;
    }
}
uint32_t crc_update(uint32_t c, const uint8_t *p, size_t n)
{
    for (size_t i = 0; i < n; ++i) c = g_crc_table[(c ^ p[i]) & 0xff] ^ (c >> 8);
    return c;
}
void put_be32(std::vector<uint8_t> &v, uint32_t x)
{
    v.push_back(x >> 24), v.push_back(x >> 16), v.push_back(x >> 8), v.push_back(x);
}
void put_chunk(std::vector<uint8_t> &out, const char *type, const std::vector<uint8_t> &data)
{
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    put_be32(out, crc_update(0xffffffffu, out.data() + start, out.size() - start) ^ 0xffffffffu);
}

// pixels: ny rows of nx words 0xAABBGGRR, i.e. bytes R G B A in memory on this little-endian host -- the
// buffer the reference hands to stbi_write_png(..., 4 components, stride 4*nx) at src/main.cpp:755-759
bool write_png(const std::string &path, int nx, int ny, const uint32_t *pixels)
{
    crc_init();
    std::vector<uint8_t> raw;
    raw.reserve((size_t)ny * (4 * (size_t)nx + 1));
    for (int j = 0; j < ny; ++j) {
        raw.push_back(0); // filter: none
        const uint8_t *row = reinterpret_cast<const uint8_t *>(pixels + (size_t)j * nx);
        raw.insert(raw.end(), row, row + 4 * (size_t)nx);
    }
    std::vector<uint8_t> z;
    z.push_back(0x78), z.push_back(0x01);
    uint32_t s1 = 1, s2 = 0;
    for (size_t pos = 0; pos < raw.size() || pos == 0;) {
        size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0); // BFINAL, BTYPE = 00
        z.push_back(n & 0xff), z.push_back(n >> 8), z.push_back(~n & 0xff), z.push_back((~n >> 8) & 0xff);
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t i = 0; i < n; ++i) {
            s1 = (s1 + raw[pos + i]) % 65521u;
            s2 = (s2 + s1) % 65521u;
        }
        pos += n;
        if (n == 0) break;
    }
    put_be32(z, (s2 << 16) | s1);

    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)nx), put_be32(ihdr, (uint32_t)ny);
    ihdr.push_back(8), ihdr.push_back(6), ihdr.push_back(0), ihdr.push_back(0), ihdr.push_back(0);
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    return fclose(f) == 0 && ok;
}

uint32_t pack(float r, float g, float b)
{ // src/main.cpp:736-741
    unsigned int ir = (unsigned int)(r * 255.99f), ig = (unsigned int)(g * 255.99f), ib = (unsigned int)(b * 255.99f);
    return 0xff000000u | (ib << 16) | (ig << 8) | (ir << 0);
}

void check(int rc, const char *what)
{
    if (rc != RT_OK) die(std::string(what) + ": " + rt_last_error());
}

std::string base_name(const std::string &out) { return out.substr(0, out.find_last_of(".")); }

// ---------------------------------------------------------------- accumulator file (--accum)
// The sums of rt_render ([A, B, G, R] per pixel, A = samples) with what they were rendered with.  Sample indices
// [0, samples) are in the file; a resumed run renders [samples, -s) with the same seed, so the image does not depend
// on where a run was interrupted (only float summation order does).
struct AccumHeader {
    char magic[8]; // "RTACC1\0\0"
    int32_t nx, ny, samples, sub_samples, max_depth;
    uint32_t seed;
    char scene[40];
};
AccumHeader accum_header(int nx, int ny, int samples, int nss, int depth, uint32_t seed, const std::string &scene)
{
    AccumHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "RTACC1", 6);
    h.nx = nx, h.ny = ny, h.samples = samples, h.sub_samples = nss, h.max_depth = depth, h.seed = seed;
    strncpy(h.scene, scene.c_str(), sizeof h.scene - 1);
    return h;
}
// returns the number of samples in the file (0 if there is no file); dies if the file is for another render
int load_accum(const std::string &path, const AccumHeader &want, std::vector<float> &sums)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    AccumHeader h;
    bool ok = fread(&h, sizeof h, 1, f) == 1 && !memcmp(h.magic, want.magic, 8) && h.nx == want.nx && h.ny == want.ny &&
              h.sub_samples == want.sub_samples && h.max_depth == want.max_depth && h.seed == want.seed &&
              !strncmp(h.scene, want.scene, sizeof h.scene) && h.samples >= 0;
    ok = ok && fread(sums.data(), sizeof(float), sums.size(), f) == sums.size();
    fclose(f);
    if (!ok) die("accumulator file " + path + " is not from this render (scene, size, -S, -r, --seed must match)");
    return h.samples;
}
void save_accum(const std::string &path, AccumHeader h, int samples, const std::vector<float> &sums)
{
    h.samples = samples;
    const std::string tmp = path + ".tmp"; // written aside and renamed: an interrupted save leaves the old file whole
    FILE *f = fopen(tmp.c_str(), "wb");
    bool ok = f && fwrite(&h, sizeof h, 1, f) == 1 && fwrite(sums.data(), sizeof(float), sums.size(), f) == sums.size();
    ok = (f ? fclose(f) == 0 : false) && ok;
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) die("cannot write " + path);
}

} // namespace

int main(int argc, char **argv)
{
    parse_args(argc, argv);
    const int nx = opt_int("w"), ny = opt_int("h"), ns = opt_int("s"), nss = opt_int("S"), bounces = opt_int("r");
    const int threads = opt_int("t"), windowed = opt_int("windowed"), passes = opt_int("p"), separate = opt_int("m");
    const int dontrender = opt_int("x"), verbose = opt_int("v"), extraverbose = opt_int("V");
    const std::string out_filename = opt_str("o");

    if (verbose) { // src/main.cpp:419-444
        printf("\n");
        if (extraverbose) printf("~ raytracer by nfauvet ~\n\n");
        printf("Resolution          : %dx%d\n", nx, ny);
        printf("Samples per pixel   : %d\n", ns);
        printf("Samples per thread  : %d\n", nss);
        printf("Bounce depth        : %d\n", bounces);
        printf("Number of Threads   : %d\n", threads);
        printf("Windowed            : %d\n", windowed);
        printf("Input file          : \"%s\"\n", opt_str("i").c_str());
        printf("Output file         : \"%s\"\n", out_filename.c_str());
        printf("Render passes       : %d\n", passes);
        printf("One file per sample : %d\n", separate);
        printf("ROI x               : %d\n", opt_int("rx"));
        printf("ROI y               : %d\n", opt_int("ry"));
        printf("ROI width           : %d\n", opt_int("rw"));
        printf("ROI height          : %d\n", opt_int("rh"));
        printf("Exit without render : %d\n", dontrender);
        printf("Verbose             : %d\n", verbose);
        printf("Extra verbose       : %d\n", extraverbose);
        // src/main.cpp:446-477: the reference counts the buffers IT allocates (one float image per sample), in
        // 32-bit arithmetic; the same figure is printed so the text matches.  This program holds one float4
        // image on the device and one on the host, whatever -s is.
        uint32_t mem = 0;
        mem += (uint32_t)((size_t)(nx * ny) * sizeof(unsigned int));
        mem += (uint32_t)((size_t)(int)((uint32_t)ns * 4u * (uint32_t)nx * (uint32_t)ny) * sizeof(float));
        mem += (uint32_t)((size_t)(3 * nx * ny) * sizeof(float));
        mem += (uint32_t)((size_t)(nx * ny) * sizeof(float));
        mem += (uint32_t)((size_t)(2 * nx * ny) * sizeof(float));
        if (mem > 1024u * 1024u * 1024u)
            printf("Buffer memory       : %u GiB\n", mem / (1024u * 1024u * 1024u));
        else if (mem > 1024u * 1024u)
            printf("Buffer memory       : %u MiB\n", mem / (1024u * 1024u));
        else if (mem > 1024u)
            printf("Buffer memory       : %u KiB\n", mem / 1024u);
        else
            printf("Buffer memory       : %u Bytes\n", mem);
    }
    if (dontrender) { // src/main.cpp:479-483
        printf("Early exit.\n");
        return 0;
    }
    if (nx <= 0 || ny <= 0 || ns <= 0 || nss <= 0) die("width, height, samples and sub-samples must be positive");

    const size_t npix = (size_t)nx * ny;
    std::vector<uint32_t> image(npix);
    std::vector<float> accum(4 * npix);
    const std::string accum_file = opt_str("accum");
    const int save_every = opt_int("save-every");
    const AccumHeader acc_head = accum_header(nx, ny, 0, nss, bounces, (uint32_t)opt_int("seed"), opt_str("scene"));
    int samples_done = 0;
    if (!accum_file.empty()) {
        samples_done = load_accum(accum_file, acc_head, accum);
        if (samples_done > 0) printf("Resuming at sample %d of %d from %s\n", samples_done, ns, accum_file.c_str());
        if (separate) die("--accum and -m do not go together");
    } else if (save_every) {
        die("--save-every needs --accum");
    }

    // --devices: "0-7", "0,2,5", "3"
    std::vector<int32_t> devices;
    if (!opt_str("devices").empty()) {
        const std::string &d = opt_str("devices");
        size_t at = 0;
        while (at < d.size()) {
            char *end = nullptr;
            long a = strtol(d.c_str() + at, &end, 10), b = a;
            if (end == d.c_str() + at || a < 0) die("--devices takes a list like 0-7 or 0,2,5");
            at = (size_t)(end - d.c_str());
            if (at < d.size() && d[at] == '-') {
                b = strtol(d.c_str() + at + 1, &end, 10);
                if (end == d.c_str() + at + 1 || b < a) die("--devices takes a list like 0-7 or 0,2,5");
                at = (size_t)(end - d.c_str());
            }
            for (long k = a; k <= b && devices.size() < 64; ++k) devices.push_back((int32_t)k);
            if (at < d.size()) {
                if (d[at] != ',') die("--devices takes a list like 0-7 or 0,2,5");
                ++at;
            }
        }
        if (devices.empty()) die("--devices names no device");
        if (separate || !accum_file.empty() || passes) die("--devices does not go together with -m, -p or --accum");
    }

    // scene and camera setup, src/main.cpp:499-511
    rt_host_scene *host = nullptr;
    check(rt_host_scene_build(opt_str("scene").c_str(), float(nx) / float(ny), &host), "scene");
    rt_scene *scene = nullptr;
    if (devices.empty()) check(rt_scene_create(rt_host_scene_desc(host), opt_int("device"), &scene), "rt_scene_create");
    const std::string &path = opt_str("path");
    int path_id = path == "auto" ? RT_PATH_AUTO : path == "exact" ? RT_PATH_EXACT : path == "fast" ? RT_PATH_FAST : -1;
    if (path_id < 0) die("--path takes auto, exact or fast");
    if (scene) check(rt_scene_set_option(scene, RT_OPT_RENDER_PATH, path_id), "--path");
    if (!devices.empty() && path_id != RT_PATH_AUTO) die("--devices renders on the path the library picks (--path auto)");

    auto time_start = std::chrono::high_resolution_clock::now();

    // the pre-pass, src/main.cpp:527-540 (a no-op in the reference; the passes are only computed when asked for)
    std::vector<rt_hit> first_hits;
    if (passes) {
        std::vector<float> st(2 * npix), rays(7 * npix);
        for (int row = 0; row < ny; ++row) {
            int j = ny - 1 - row; // src/main.cpp:272-286
            for (int i = 0; i < nx; ++i) {
                st[2 * ((size_t)row * nx + i) + 0] = float(i + 0.5f) / float(nx);
                st[2 * ((size_t)row * nx + i) + 1] = float(j + 0.5f) / float(ny);
            }
        }
        first_hits.resize(npix);
        check(rt_camera_rays(scene, (int32_t)npix, st.data(), nullptr, nullptr, rays.data()), "rt_camera_rays");
        check(rt_trace_rays(scene, (int32_t)npix, rays.data(), 0.001f, 3.402823466e+38f, (uint32_t)opt_int("seed"),
                            first_hits.data()),
              "rt_trace_rays");
    }
    printf("Single pass finished\n"); // src/main.cpp:265

    rt_render_params rp;
    memset(&rp, 0, sizeof rp);
    rp.struct_size = sizeof rp;
    rp.nx = nx, rp.ny = ny;
    rp.sub_samples = nss;
    rp.max_depth = bounces;
    rp.seed = (uint32_t)opt_int("seed");

    std::vector<std::string> sample_files;
    if (separate) {
        // one image per sample (src/main.cpp:663-706): each sample index is rendered by its own call, written,
        // and added to the total on the host in sample order -- the order of the reference's resolve loop
        std::vector<float> one(4 * npix);
        std::fill(accum.begin(), accum.end(), 0.0f);
        for (int s = 0; s < ns; ++s) {
            rp.sample_begin = s, rp.sample_end = s + 1;
            check(rt_render_to_host(scene, &rp, one.data(), image.data()), "rt_render");
            for (size_t k = 0; k < 4 * npix; ++k) accum[k] += one[k];
            printf("%d/%d : %.2f%%\r", s + 1, ns, 100.0f * (float)(s + 1) / (float)ns);
            fflush(stdout);
            std::string name = base_name(out_filename) + "_" + std::to_string(s) + ".png";
            if (!write_png(name, nx, ny, image.data())) die("cannot write " + name);
            sample_files.push_back(name);
        }
        float one_over_n_samples = 1.0f / ns; // src/main.cpp:709-742
        for (size_t k = 0; k < npix; ++k) {
            float c[3] = {accum[4 * k + 3] * one_over_n_samples, accum[4 * k + 2] * one_over_n_samples,
                          accum[4 * k + 1] * one_over_n_samples};
            for (float &x : c) {
                x = sqrtf(x);
                x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
            }
            image[k] = pack(c[0], c[1], c[2]);
        }
    } else if (!accum_file.empty()) {
        // progressive: sample ranges [done, done + step), sums added on the host, file saved after each range
        std::vector<float> part(4 * npix);
        if (samples_done == 0) std::fill(accum.begin(), accum.end(), 0.0f);
        while (samples_done < ns) {
            const int step = save_every > 0 ? std::min(save_every, ns - samples_done) : ns - samples_done;
            rp.sample_begin = samples_done, rp.sample_end = samples_done + step;
            check(rt_render_to_host(scene, &rp, part.data(), nullptr), "rt_render");
            for (size_t k = 0; k < 4 * npix; ++k) accum[k] += part[k];
            samples_done += step;
            save_accum(accum_file, acc_head, samples_done, accum);
            printf("%d/%d : %.2f%%\r", samples_done, ns, 100.0f * (float)samples_done / (float)ns);
            fflush(stdout);
        }
        for (size_t k = 0; k < npix; ++k) { // src/main.cpp:709-742 with the number of samples in the sums
            const float inv = 1.0f / accum[4 * k];
            float c[3] = {accum[4 * k + 3] * inv, accum[4 * k + 2] * inv, accum[4 * k + 1] * inv};
            for (float &x : c) {
                x = sqrtf(x);
                x = x < 0.0f ? 0.0f : (x > 1.0f ? 1.0f : x);
            }
            image[k] = pack(c[0], c[1], c[2]);
        }
    } else if (!devices.empty()) {
        // the samples shared by several GPUs: disjoint sample ranges, one RCCL sum-reduce, resolve on the first (include/rt_amd.h).
        // The context -- scenes, streams, accumulators, ONE communicator -- is built once, like the reference's thread pool
        // (src/main.cpp:523-561); --repeat K renders the image K times on it and prints what each one took, so that a
        // seconds-to-image is render + reduce + resolve + copy and not communicator set-up.
        rp.sample_begin = 0, rp.sample_end = ns;
        rt_multi *ctx = nullptr;
        auto t_ctx = std::chrono::high_resolution_clock::now();
        check(rt_multi_create(rt_host_scene_desc(host), devices.data(), (int32_t)devices.size(),
                              opt_int("reduce-always") ? RT_MULTI_REDUCE_ALWAYS : 0u, &ctx),
              "rt_multi_create");
        const int repeat = std::max(1, opt_int("repeat"));
        if (repeat > 1)
            printf("context on %zu device(s): %.2fms\n", devices.size(),
                   std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_ctx).count());
        for (int k = 0; k < repeat; ++k) {
            auto t_img = std::chrono::high_resolution_clock::now();
            check(rt_multi_render_to_host(ctx, &rp, accum.data(), image.data()), "rt_multi_render_to_host");
            if (repeat > 1) {
                float kernel_ms = 0.0f;
                check(rt_multi_last_render_ms(ctx, &kernel_ms), "rt_multi_last_render_ms");
                printf("image %d/%d: %.2fms (slowest device's render kernel %.2fms)\n", k + 1, repeat,
                       std::chrono::duration<double, std::milli>(std::chrono::high_resolution_clock::now() - t_img).count(), kernel_ms);
            }
        }
        rt_multi_destroy(ctx);
        printf("%d/%d : %.2f%%\r", ns, ns, 100.0f);
    } else {
        rp.sample_begin = 0, rp.sample_end = ns;
        check(rt_render_to_host(scene, &rp, accum.data(), image.data()), "rt_render");
        printf("%d/%d : %.2f%%\r", ns, ns, 100.0f); // src/main.cpp:213-214, once: the samples finish together
    }
    printf("\n"); // src/main.cpp:563

    if (passes) { // src/main.cpp:566-660, fed by the pre-pass above
        const float *o = rt_host_scene_desc(host)->camera.origin;
        std::vector<float> depth(npix, 0.0f);
        float depth_min = 0.0f, depth_max = 0.0f;
        std::vector<uint32_t> aov(npix);
        for (size_t k = 0; k < npix; ++k) {
            const rt_hit &h = first_hits[k];
            float n[3] = {0, 0, 0};
            if (h.prim_id >= 0) {
                float len = sqrtf(h.normal[0] * h.normal[0] + h.normal[1] * h.normal[1] + h.normal[2] * h.normal[2]);
                for (int c = 0; c < 3; ++c) n[c] = h.normal[c] / len;
                float d[3] = {h.p[0] - o[0], h.p[1] - o[1], h.p[2] - o[2]};
                depth[k] = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                if (depth[k] < depth_min) depth_min = depth[k];
                if (depth[k] > depth_max) depth_max = depth[k];
            }
            aov[k] = pack(0.5f * (n[0] + 1.0f), 0.5f * (n[1] + 1.0f), 0.5f * (n[2] + 1.0f));
        }
        std::string name = base_name(out_filename) + "_NORMAL.png";
        printf("output: %s\n", name.c_str());
        if (!write_png(name, nx, ny, aov.data())) die("cannot write " + name);
        for (size_t k = 0; k < npix; ++k) {
            float d = depth_max > depth_min ? (depth[k] - depth_min) / (depth_max - depth_min) : 0.0f;
            aov[k] = pack(d, d, d);
        }
        name = base_name(out_filename) + "_DEPTH.png";
        printf("output: %s\n", name.c_str());
        if (!write_png(name, nx, ny, aov.data())) die("cannot write " + name);
        for (size_t k = 0; k < npix; ++k) {
            const rt_hit &h = first_hits[k];
            float u = h.prim_id >= 0 ? h.u : 0.0f, v = h.prim_id >= 0 ? h.v : 0.0f;
            u = u < 0.0f ? 0.0f : (u > 1.0f ? 1.0f : u); // moving spheres leave (u,v) unset in the reference
            v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
            aov[k] = pack(u, v, 0.0f);
        }
        name = base_name(out_filename) + "_UV.png";
        printf("output: %s\n", name.c_str());
        if (!write_png(name, nx, ny, aov.data())) die("cannot write " + name);
    }

    for (const std::string &name : sample_files) printf("output: %s\n", name.c_str()); // src/main.cpp:699

    auto time_end = std::chrono::high_resolution_clock::now();
    printf("Time: %.2fms\n", std::chrono::duration<double, std::milli>(time_end - time_start).count());
    printf("Writing file.\n");
    if (!write_png(out_filename, nx, ny, image.data())) die("cannot write " + out_filename);

    if (scene) rt_scene_destroy(scene);
    rt_host_scene_free(host);
    return 0;
}
