// the I/O edges of the texturing path in native code, so the per-shape host work (Python `%f` formatting of
// ~5 k vertices + 10 k faces, 8 x 4 PNG encodes, PLY parsing) does not cap shapes/hour once the GPU work is sub-second.
//   pdhip_io_ply_count / pdhip_io_read_ply_xyzrgb   utils/other_utils.py:155-163 (plyfile): x,y,z + red,green,blue of the
//                                                    `vertex` element, binary_little_endian or ascii
//   pdhip_io_write_obj_mtl                           models/get3d/get3d_utils/utils_3d.py:27-64 (savemeshtes2), byte-identical text
//   pdhip_io_write_png                               utils/utils_2d.py:351-399 (PIL save): 8-bit RGB / RGBA, zlib deflate
//   pdhip_chw_f32_to_hwc_u8 (device)                 the `(img * 255).clip(0, 255).astype(uint8)` + CHW->HWC step before the save
// Host code apart from the last one; no HIP call is made by the file functions.
#include "common.h"
#include <zlib.h>
#include <string>
#include <vector>
#include <thread>
#include <atomic>
#include <sstream>
#include <string.h>
using namespace pdhip;

namespace {
struct PlyProp { std::string name; int bytes; char kind; };     // kind: f float, d double, u unsigned, i signed
struct PlyHeader { long long n = 0; bool binary = false; std::vector<PlyProp> props; long data_offset = 0; };

bool ply_type(const std::string& t, PlyProp* p) {
    if (t == "float" || t == "float32") { p->bytes = 4; p->kind = 'f'; }
    else if (t == "double" || t == "float64") { p->bytes = 8; p->kind = 'd'; }
    else if (t == "uchar" || t == "uint8") { p->bytes = 1; p->kind = 'u'; }
    else if (t == "char" || t == "int8") { p->bytes = 1; p->kind = 'i'; }
    else if (t == "ushort" || t == "uint16") { p->bytes = 2; p->kind = 'u'; }
    else if (t == "short" || t == "int16") { p->bytes = 2; p->kind = 'i'; }
    else if (t == "uint" || t == "uint32") { p->bytes = 4; p->kind = 'u'; }
    else if (t == "int" || t == "int32") { p->bytes = 4; p->kind = 'i'; }
    else return false;
    return true;
}

int read_header(FILE* f, PlyHeader* h) {
    char line[1024];
    bool in_vertex = false, seen_end = false, first = true;
    while (fgets(line, sizeof line, f)) {
        std::istringstream ss(line);
        std::string a, b, c;
        ss >> a >> b >> c;
        if (first) { if (a != "ply") return -1; first = false; continue; }
        if (a == "format") h->binary = b == "binary_little_endian";
        if (a == "format" && b != "binary_little_endian" && b != "ascii") return -2;
        if (a == "element") { in_vertex = b == "vertex"; if (in_vertex) h->n = atoll(c.c_str()); }
        if (a == "property" && in_vertex) {
            if (b == "list") return -3;
            PlyProp p; p.name = c;
            if (!ply_type(b, &p)) return -3;
            h->props.push_back(p);
        }
        if (a == "end_header") { seen_end = true; break; }
    }
    if (!seen_end) return -1;
    h->data_offset = ftell(f);
    return 0;
}

double load_scalar(const unsigned char* p, const PlyProp& pr) {
    switch (pr.kind) {
        case 'f': { float v; memcpy(&v, p, 4); return v; }
        case 'd': { double v; memcpy(&v, p, 8); return v; }
        case 'u': if (pr.bytes == 1) return *p; if (pr.bytes == 2) { uint16_t v; memcpy(&v, p, 2); return v; } { uint32_t v; memcpy(&v, p, 4); return v; }
        default: if (pr.bytes == 1) return (int8_t)*p; if (pr.bytes == 2) { int16_t v; memcpy(&v, p, 2); return v; } { int32_t v; memcpy(&v, p, 4); return v; }
    }
}
}  // namespace

extern "C" long long pdhip_io_ply_count(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { set_error("pdhip_io_ply_count: cannot open %s", path); return -1; }
    PlyHeader h;
    const int rc = read_header(f, &h);
    fclose(f);
    if (rc) { set_error("pdhip_io_ply_count: bad or unsupported PLY header in %s (%d)", path, rc); return -1; }
    return h.n;
}

extern "C" int pdhip_io_read_ply_xyzrgb(const char* path, float* xyz /*[n,3]*/, uint8_t* rgb /*[n,3]*/, long long n) {
    PD_REQUIRE(path && xyz && rgb && n >= 0, "pdhip_io_read_ply_xyzrgb: bad arguments");
    FILE* f = fopen(path, "rb");
    PD_REQUIRE(f != nullptr, "pdhip_io_read_ply_xyzrgb: cannot open %s", path);
    PlyHeader h;
    int rc = read_header(f, &h);
    if (rc || h.n != n) { fclose(f); set_error("pdhip_io_read_ply_xyzrgb: bad header or vertex count mismatch in %s", path); return PDHIP_E_ARG; }
    int col[6] = {-1, -1, -1, -1, -1, -1};
    const char* want[6] = {"x", "y", "z", "red", "green", "blue"};
    std::vector<int> offs(h.props.size());
    int stride = 0;
    for (size_t i = 0; i < h.props.size(); ++i) {
        offs[i] = stride; stride += h.props[i].bytes;
        for (int k = 0; k < 6; ++k) if (h.props[i].name == want[k]) col[k] = (int)i;
    }
    for (int k = 0; k < 6; ++k)
        if (col[k] < 0) { fclose(f); set_error("pdhip_io_read_ply_xyzrgb: property %s missing in %s", want[k], path); return PDHIP_E_ARG; }
    if (h.binary) {
        std::vector<unsigned char> buf((size_t)stride * (size_t)n);
        const size_t got = fread(buf.data(), 1, buf.size(), f);
        fclose(f);
        PD_REQUIRE(got == buf.size(), "pdhip_io_read_ply_xyzrgb: %s is truncated", path);
        for (long long i = 0; i < n; ++i) {
            const unsigned char* r = buf.data() + (size_t)i * stride;
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = (float)load_scalar(r + offs[col[k]], h.props[col[k]]);
            for (int k = 0; k < 3; ++k) rgb[3 * i + k] = (uint8_t)load_scalar(r + offs[col[3 + k]], h.props[col[3 + k]]);
        }
    } else {
        std::vector<double> row(h.props.size());
        for (long long i = 0; i < n; ++i) {
            for (size_t p = 0; p < h.props.size(); ++p)
                if (fscanf(f, "%lf", &row[p]) != 1) { fclose(f); set_error("pdhip_io_read_ply_xyzrgb: %s: bad ascii row %lld", path, i); return PDHIP_E_ARG; }
            for (int k = 0; k < 3; ++k) xyz[3 * i + k] = (float)row[col[k]];
            for (int k = 0; k < 3; ++k) rgb[3 * i + k] = (uint8_t)row[col[3 + k]];
        }
        fclose(f);
    }
    return PDHIP_OK;
}

// OBJ + MTL exactly as savemeshtes2 writes them: `v %f %f %f`, `vt %f %f`, `f a/b c/d e/f` (1-based), material block.
extern "C" int pdhip_io_write_obj_mtl(const char* obj_path, const char* mtl_path, const char* texture_stem, const double* points,
                                      long long P, const double* tcoords, long long T, const int64_t* faces, const int64_t* facetex,
                                      long long F) {
    PD_REQUIRE(obj_path && mtl_path && texture_stem && points && tcoords && faces && facetex, "pdhip_io_write_obj_mtl: null argument");
    FILE* m = fopen(mtl_path, "w");
    PD_REQUIRE(m != nullptr, "pdhip_io_write_obj_mtl: cannot open %s", mtl_path);
    fprintf(m, "newmtl material_0\nKd 1 1 1\nKa 0 0 0\nKs 0.4 0.4 0.4\nNs 10\nillum 2\nmap_Kd %s.png\n", texture_stem);
    fclose(m);
    std::string out;
    out.reserve((size_t)(P * 40 + T * 28 + F * 48 + 64));
    char buf[256];
    out += "mtllib "; out += texture_stem; out += ".mtl\n";
    for (long long i = 0; i < P; ++i) {
        const int k = snprintf(buf, sizeof buf, "v %f %f %f\n", points[3 * i], points[3 * i + 1], points[3 * i + 2]);
        out.append(buf, k);
    }
    for (long long i = 0; i < T; ++i) {
        const int k = snprintf(buf, sizeof buf, "vt %f %f\n", tcoords[2 * i], tcoords[2 * i + 1]);
        out.append(buf, k);
    }
    out += "usemtl material_0\n";
    for (long long i = 0; i < F; ++i) {
        const int k = snprintf(buf, sizeof buf, "f %lld/%lld %lld/%lld %lld/%lld\n", (long long)faces[3 * i] + 1, (long long)facetex[3 * i] + 1,
                               (long long)faces[3 * i + 1] + 1, (long long)facetex[3 * i + 1] + 1, (long long)faces[3 * i + 2] + 1,
                               (long long)facetex[3 * i + 2] + 1);
        out.append(buf, k);
    }
    FILE* f = fopen(obj_path, "w");
    PD_REQUIRE(f != nullptr, "pdhip_io_write_obj_mtl: cannot open %s", obj_path);
    const size_t w = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    PD_REQUIRE(w == out.size(), "pdhip_io_write_obj_mtl: short write to %s", obj_path);
    return PDHIP_OK;
}

namespace {
void put_chunk(std::vector<unsigned char>& png, const char* type, const unsigned char* data, size_t len) {
    const uint32_t L = (uint32_t)len;
    const unsigned char lb[4] = {(unsigned char)(L >> 24), (unsigned char)(L >> 16), (unsigned char)(L >> 8), (unsigned char)L};
    png.insert(png.end(), lb, lb + 4);
    const size_t at = png.size();
    png.insert(png.end(), type, type + 4);
    if (len) png.insert(png.end(), data, data + len);
    const uint32_t c = (uint32_t)crc32(0L, png.data() + at, (uInt)(len + 4));
    const unsigned char cb[4] = {(unsigned char)(c >> 24), (unsigned char)(c >> 16), (unsigned char)(c >> 8), (unsigned char)c};
    png.insert(png.end(), cb, cb + 4);
}
}  // namespace

// 8-bit RGB (channels 3) or RGBA (4) PNG, filter type 0 rows, one zlib stream
extern "C" int pdhip_io_write_png(const char* path, const uint8_t* hwc, int H, int W, int channels, int level) {
    PD_REQUIRE(path && hwc && H > 0 && W > 0 && (channels == 3 || channels == 4), "pdhip_io_write_png: bad arguments");
    const size_t row = (size_t)W * channels;
    std::vector<unsigned char> raw((row + 1) * H);
    for (int y = 0; y < H; ++y) {
        raw[(row + 1) * y] = 0;
        memcpy(&raw[(row + 1) * y + 1], hwc + row * y, row);
    }
    std::vector<unsigned char> comp;
    uLongf clen = 0;
    const int lvl = level < 0 ? 1 : level;
    const size_t STRIP = 256 * 1024;
    if (raw.size() >= 2 * STRIP) {
        // Large images (the 1024^2 atlas: 3-4 MB) are deflated in parallel strips, pigz style: every strip is a raw deflate stream
        // closed with a sync flush (byte-aligned, empty stored block), the last one with Z_FINISH; concatenated behind one zlib
        // header and followed by the Adler-32 of the whole image they form one valid zlib stream.  60 ms -> a few ms per atlas.
        const int ns = (int)((raw.size() + STRIP - 1) / STRIP);
        std::vector<std::vector<unsigned char>> part(ns);
        std::vector<uLong> adl(ns);
        std::vector<int> ok(ns, 0);
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::atomic<int> next{0};
        auto work = [&]() {
            for (int i = next.fetch_add(1); i < ns; i = next.fetch_add(1)) {
                const size_t b = (size_t)i * STRIP, n = std::min(STRIP, raw.size() - b);
                z_stream z;
                memset(&z, 0, sizeof(z));
                if (deflateInit2(&z, lvl, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) continue;
                part[i].resize(deflateBound(&z, (uLong)n) + 16);
                z.next_in = raw.data() + b; z.avail_in = (uInt)n;
                z.next_out = part[i].data(); z.avail_out = (uInt)part[i].size();
                const int rc = deflate(&z, i == ns - 1 ? Z_FINISH : Z_SYNC_FLUSH);
                ok[i] = (i == ns - 1 ? rc == Z_STREAM_END : rc == Z_OK) && z.avail_in == 0;
                part[i].resize(part[i].size() - z.avail_out);
                deflateEnd(&z);
                adl[i] = adler32(adler32(0L, Z_NULL, 0), raw.data() + b, (uInt)n);
            }
        };
        std::vector<std::thread> th;
        for (unsigned t = 1; t < std::min<unsigned>(hw, (unsigned)ns); ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
        uLong ad = adler32(0L, Z_NULL, 0);
        comp.push_back(0x78); comp.push_back(0x01);
        for (int i = 0; i < ns; ++i) {
            PD_REQUIRE(ok[i], "pdhip_io_write_png: deflate failed");
            comp.insert(comp.end(), part[i].begin(), part[i].end());
            const size_t b = (size_t)i * STRIP, n = std::min(STRIP, raw.size() - b);
            ad = i == 0 ? adl[0] : adler32_combine(ad, adl[i], (z_off_t)n);
        }
        const unsigned char ab[4] = {(unsigned char)(ad >> 24), (unsigned char)(ad >> 16), (unsigned char)(ad >> 8), (unsigned char)ad};
        comp.insert(comp.end(), ab, ab + 4);
        clen = (uLongf)comp.size();
    } else {
        clen = compressBound((uLong)raw.size());
        comp.resize(clen);
        PD_REQUIRE(compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), lvl) == Z_OK, "pdhip_io_write_png: deflate failed");
    }
    std::vector<unsigned char> png;
    const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    png.insert(png.end(), sig, sig + 8);
    unsigned char ihdr[13] = {(unsigned char)(W >> 24), (unsigned char)(W >> 16), (unsigned char)(W >> 8), (unsigned char)W,
                              (unsigned char)(H >> 24), (unsigned char)(H >> 16), (unsigned char)(H >> 8), (unsigned char)H,
                              8, (unsigned char)(channels == 3 ? 2 : 6), 0, 0, 0};
    put_chunk(png, "IHDR", ihdr, 13);
    put_chunk(png, "IDAT", comp.data(), clen);
    put_chunk(png, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    PD_REQUIRE(f != nullptr, "pdhip_io_write_png: cannot open %s", path);
    const size_t w = fwrite(png.data(), 1, png.size(), f);
    fclose(f);
    PD_REQUIRE(w == png.size(), "pdhip_io_write_png: short write to %s", path);
    return PDHIP_OK;
}

// uint8(clip(img * 255, 0, 255)) with CHW -> HWC, on the device (so only H*W*C bytes cross PCIe)
__global__ void k_chw_to_hwc_u8(const float* __restrict__ img, int C, long long HW, uint8_t* __restrict__ out) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < HW; p += (long long)gridDim.x * blockDim.x)
        for (int c = 0; c < C; ++c) {
            float v = img[(size_t)c * HW + p] * 255.0f;
            v = fminf(fmaxf(v, 0.0f), 255.0f);
            out[(size_t)p * C + c] = (uint8_t)v;
        }
}

extern "C" int pdhip_chw_f32_to_hwc_u8(const float* img, int C, int H, int W, uint8_t* out, void* stream) {
    PD_REQUIRE(img && out && C > 0 && C <= 4 && H > 0 && W > 0, "pdhip_chw_f32_to_hwc_u8: bad arguments");
    const long long HW = (long long)H * W;
    k_chw_to_hwc_u8<<<min(cdiv(HW, 256), 4096), 256, 0, as_stream(stream)>>>(img, C, HW, out);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
