// Host side of the drop-in (include/rtamd_host.h): YAML scene subset, OBJ import,
// PNG I/O, camera, sampling parser and the compute_image driver, restating the
// reference's behaviour (file:line relative to /root/reference/src/).
#include <zlib.h>

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rtamd_host.h"
#include "bvh.h"

namespace {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
[[noreturn]] void raise(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    throw Error(buf);
}
void set_err(char* err, size_t n, const std::string& m) {
    if (err && n) snprintf(err, n, "%s", m.c_str());
}

// ------------------------------------------------------------------ YAML subset
// Block maps, block sequences (of maps or scalars), flow sequences of scalars, plain and
// quoted scalars, comments.  That is everything serde_yaml sees in the reference's scene
// files.
struct Node {
    enum Kind { Null, Scalar, Seq, Map } kind = Null;
    std::string scalar;
    std::vector<Node> seq;
    std::vector<std::pair<std::string, Node>> map;
    int line = 0;
    const Node* get(const std::string& k) const {
        if (kind != Map) return nullptr;
        for (auto& kv : map) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

struct Line {
    int indent;
    std::string text;
    int no;
};

std::string trim(const std::string& s) {
    size_t a = 0, b = s.size();
    while (a < b && (s[a] == ' ' || s[a] == '\t' || s[a] == '\r')) a++;
    while (b > a && (s[b - 1] == ' ' || s[b - 1] == '\t' || s[b - 1] == '\r')) b--;
    return s.substr(a, b - a);
}
std::string strip_comment(const std::string& s) {
    bool sq = false, dq = false;
    for (size_t i = 0; i < s.size(); ++i) {
        char c = s[i];
        if (c == '\'' && !dq) sq = !sq;
        else if (c == '"' && !sq) dq = !dq;
        else if (c == '#' && !sq && !dq && (i == 0 || s[i - 1] == ' ' || s[i - 1] == '\t')) return s.substr(0, i);
    }
    return s;
}
std::string unquote(const std::string& s) {
    if (s.size() >= 2 && ((s.front() == '"' && s.back() == '"') || (s.front() == '\'' && s.back() == '\''))) return s.substr(1, s.size() - 2);
    return s;
}

struct YamlParser {
    std::vector<Line> lines;
    size_t pos = 0;

    explicit YamlParser(std::istream& in) {
        std::string l;
        int no = 0;
        while (std::getline(in, l)) {
            ++no;
            std::string s = strip_comment(l);
            std::string t = trim(s);
            if (t.empty() || t == "---") continue;
            int ind = 0;
            while (ind < (int)s.size() && s[ind] == ' ') ind++;
            if (ind < (int)s.size() && s[ind] == '\t') raise("yaml line %d: tabs are not allowed for indentation", no);
            lines.push_back({ind, t, no});
        }
    }

    static Node parse_flow_or_scalar(const std::string& text, int line) {
        Node n;
        n.line = line;
        if (!text.empty() && text.front() == '[') {
            if (text.back() != ']') raise("yaml line %d: unterminated flow sequence", line);
            n.kind = Node::Seq;
            std::string inner = text.substr(1, text.size() - 2);
            size_t start = 0;
            int depth = 0;
            for (size_t i = 0; i <= inner.size(); ++i) {
                if (i < inner.size() && inner[i] == '[') depth++;
                if (i < inner.size() && inner[i] == ']') depth--;
                if (i == inner.size() || (inner[i] == ',' && depth == 0)) {
                    std::string item = trim(inner.substr(start, i - start));
                    if (!item.empty()) n.seq.push_back(parse_flow_or_scalar(item, line));
                    start = i + 1;
                }
            }
            return n;
        }
        n.kind = Node::Scalar;
        n.scalar = unquote(text);
        return n;
    }

    // splits "key: value" at the first ": " / trailing ':' outside quotes; returns false if not a mapping entry
    static bool split_key(const std::string& t, std::string& key, std::string& val) {
        bool sq = false, dq = false;
        for (size_t i = 0; i < t.size(); ++i) {
            char c = t[i];
            if (c == '\'' && !dq) sq = !sq;
            else if (c == '"' && !sq) dq = !dq;
            else if (c == '[' && !sq && !dq) return false;
            else if (c == ':' && !sq && !dq && (i + 1 == t.size() || t[i + 1] == ' ')) {
                key = unquote(trim(t.substr(0, i)));
                val = trim(t.substr(i + 1));
                return true;
            }
        }
        return false;
    }

    Node parse_block(int indent) {
        Node n;
        if (pos >= lines.size()) return n;
        n.line = lines[pos].no;
        if (lines[pos].text[0] == '-' && (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
            n.kind = Node::Seq;
            while (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-' &&
                   (lines[pos].text.size() == 1 || lines[pos].text[1] == ' ')) {
                std::string rest = trim(lines[pos].text.substr(1));
                if (rest.empty()) {
                    pos++;
                    if (pos < lines.size() && lines[pos].indent > indent) n.seq.push_back(parse_block(lines[pos].indent));
                    else n.seq.emplace_back();
                } else {
                    std::string k, v;
                    if (split_key(rest, k, v)) {
                        // "- key: value": the item is a map whose keys sit at the column of `key`
                        int item_indent = indent + (int)(lines[pos].text.size() - rest.size());
                        lines[pos].indent = item_indent;
                        lines[pos].text = rest;
                        n.seq.push_back(parse_block(item_indent));
                    } else {
                        n.seq.push_back(parse_flow_or_scalar(rest, lines[pos].no));
                        pos++;
                    }
                }
            }
            return n;
        }
        n.kind = Node::Map;
        while (pos < lines.size() && lines[pos].indent == indent) {
            std::string k, v;
            if (!split_key(lines[pos].text, k, v)) raise("yaml line %d: expected 'key: value'", lines[pos].no);
            int ln = lines[pos].no;
            pos++;
            Node child;
            if (!v.empty()) child = parse_flow_or_scalar(v, ln);
            else if (pos < lines.size() && lines[pos].indent > indent) child = parse_block(lines[pos].indent);
            else if (pos < lines.size() && lines[pos].indent == indent && lines[pos].text[0] == '-') child = parse_block(indent);
            child.line = ln;
            n.map.emplace_back(k, std::move(child));
        }
        if (pos < lines.size() && lines[pos].indent > indent) raise("yaml line %d: unexpected indentation", lines[pos].no);
        return n;
    }
    Node parse() {
        if (lines.empty()) return Node();
        Node n = parse_block(lines[0].indent);
        if (pos < lines.size()) raise("yaml line %d: unexpected content", lines[pos].no);
        return n;
    }
};

double as_f64(const Node* n, const char* what) {   // serde_yaml Value::as_f64: integers and floats
    if (!n || n->kind != Node::Scalar) raise("%s: expected a number", what);
    const std::string& s = n->scalar;
    if (s == ".inf" || s == "+.inf") return INFINITY;
    if (s == "-.inf") return -INFINITY;
    if (s == ".nan") return NAN;
    char* end = nullptr;
    errno = 0;
    double v = strtod(s.c_str(), &end);
    if (s.empty() || end != s.c_str() + s.size()) raise("%s: '%s' is not a number", what, s.c_str());
    return v;
}
uint64_t as_usize(const Node* n, const char* what) {
    if (!n) raise("missing field `%s`", what);
    if (n->kind != Node::Scalar) raise("%s: expected an unsigned integer", what);
    const std::string& s = n->scalar;
    if (s.empty()) raise("%s: expected an unsigned integer", what);
    uint64_t v = 0;
    size_t i = (s[0] == '+') ? 1 : 0;
    if (i >= s.size()) raise("%s: expected an unsigned integer", what);
    for (; i < s.size(); ++i) {
        if (s[i] < '0' || s[i] > '9') raise("%s: invalid value '%s', expected an unsigned integer", what, s.c_str());
        v = v * 10 + (uint64_t)(s[i] - '0');
    }
    return v;
}
// Vector3 / Color deserialize: a sequence, elements 0..2 as_f64 (vec3.rs:164-175, image.rs:54-65)
void as_vec3(const Node* n, const char* what, double out[3]) {
    if (!n) raise("missing field `%s`", what);
    if (n->kind != Node::Seq || n->seq.size() < 3) raise("%s: expected a sequence of 3 numbers", what);
    for (int k = 0; k < 3; ++k) out[k] = as_f64(&n->seq[k], what);
}
const Node* need(const Node& m, const char* key) {
    const Node* n = m.get(key);
    if (!n) raise("missing field `%s` (line %d)", key, m.line);
    return n;
}
std::string as_str(const Node* n, const char* what) {
    if (!n || n->kind != Node::Scalar) raise("%s: expected a string", what);
    return n->scalar;
}

// ------------------------------------------------------------------ PNG (zlib based)
uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

// image.rs:166-184.  The png crate hands back the raw decoded buffer; only 8-bit RGB
// (what the reference's textures are) is accepted here.
void read_png(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h) {
    std::ifstream f(path, std::ios::binary);
    if (!f) raise("cannot open image '%s'", path.c_str());
    std::vector<uint8_t> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (buf.size() < 8 || memcmp(buf.data(), sig, 8) != 0) raise("'%s' is not a PNG file", path.c_str());
    size_t p = 8;
    std::vector<uint8_t> idat;
    int depth = 0, ctype = -1, interlace = 0;
    bool have_ihdr = false, done = false;
    while (!done && p + 12 <= buf.size()) {
        uint32_t len = be32(&buf[p]);
        if (p + 12 + (size_t)len > buf.size()) raise("'%s': truncated PNG chunk", path.c_str());
        const uint8_t* type = &buf[p + 4];
        const uint8_t* data = &buf[p + 8];
        if (crc32(crc32(0, type, 4), data, len) != be32(data + len)) raise("'%s': PNG CRC mismatch", path.c_str());
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) raise("'%s': bad IHDR", path.c_str());
            w = be32(data); h = be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) done = true;
        p += 12 + (size_t)len;
    }
    if (!have_ihdr) raise("'%s': PNG has no IHDR", path.c_str());
    if (depth != 8 || ctype != 2) raise("'%s': only 8-bit RGB PNG textures are supported (bit depth %d, colour type %d)", path.c_str(), depth, ctype);
    if (interlace) raise("'%s': interlaced PNG is not supported", path.c_str());
    if (w == 0 || h == 0 || (uint64_t)w * h > (1ull << 30)) raise("'%s': bad PNG size", path.c_str());
    const size_t stride = (size_t)w * 3;
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf out_len = raw.size();
    int zr = uncompress(raw.data(), &out_len, idat.data(), idat.size());
    if (zr != Z_OK || out_len != raw.size()) raise("'%s': PNG inflate failed (%d)", path.c_str(), zr);
    rgb.assign(stride * h, 0);
    for (uint32_t y = 0; y < h; ++y) {
        const uint8_t* in = &raw[(stride + 1) * y];
        uint8_t ft = in[0];
        in++;
        uint8_t* out = &rgb[stride * y];
        const uint8_t* up = y ? &rgb[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= 3 ? out[i - 3] : 0, b = up ? up[i] : 0, c = (up && i >= 3) ? up[i - 3] : 0;
            int pred = 0;
            switch (ft) {
            case 0: pred = 0; break;
            case 1: pred = a; break;
            case 2: pred = b; break;
            case 3: pred = (a + b) >> 1; break;
            case 4: { int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
            default: raise("'%s': bad PNG filter %d", path.c_str(), (int)ft);
            }
            out[i] = (uint8_t)(in[i] + pred);
        }
    }
}

// image.rs:149-163
// bytes_per_sample = 1: 8-bit RGB (what the reference writes); 2: 16-bit RGB, samples big-endian (PNG spec)
void write_png(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t bytes_per_sample = 1) {
    namespace fs = std::filesystem;
    fs::path pth(path);
    if (pth.has_parent_path()) {
        std::error_code ec;
        fs::create_directories(pth.parent_path(), ec);   // image.rs:155-156
        if (ec) raise("cannot create directory '%s': %s", pth.parent_path().string().c_str(), ec.message().c_str());
    }
    const size_t stride = (size_t)w * 3 * bytes_per_sample;
    std::vector<uint8_t> raw((stride + 1) * h);
    for (uint32_t y = 0; y < h; ++y) {
        raw[(stride + 1) * y] = 0;
        memcpy(&raw[(stride + 1) * y + 1], rgb + stride * y, stride);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) raise("PNG deflate failed");
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        put32(out, len);
        size_t s = out.size();
        out.insert(out.end(), type, type + 4);
        if (len) out.insert(out.end(), data, data + len);
        put32(out, (uint32_t)crc32(0, &out[s], len + 4));
    };
    std::vector<uint8_t> ihdr;
    put32(ihdr, w); put32(ihdr, h);
    ihdr.push_back((uint8_t)(8 * bytes_per_sample)); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr.data(), (uint32_t)ihdr.size());
    chunk("IDAT", comp.data(), (uint32_t)clen);
    chunk("IEND", nullptr, 0);
    std::ofstream f(path, std::ios::binary);
    if (!f) raise("cannot create '%s'", path.c_str());
    f.write((const char*)out.data(), (std::streamsize)out.size());
    if (!f) raise("cannot write '%s'", path.c_str());
}

// ------------------------------------------------------------------ scene model
struct MaterialH {
    int type = 0;
    double rgb[3] = {0, 0, 0};
    double param = 0;
    std::string texture_path;
    int texture = -1;
};

// materials.rs:18-26 (#[serde(tag = "type")]); unknown keys such as `roughness` are ignored
MaterialH parse_material(const Node& n) {
    if (n.kind != Node::Map) raise("material (line %d): expected a mapping", n.line);
    MaterialH m;
    std::string t = as_str(need(n, "type"), "material type");
    if (t == "Lambertian") { m.type = RT_MAT_LAMBERTIAN; as_vec3(n.get("albedo"), "albedo", m.rgb); }
    else if (t == "Metal") { m.type = RT_MAT_METAL; as_vec3(n.get("albedo"), "albedo", m.rgb); m.param = as_f64(need(n, "fuzziness"), "fuzziness"); }
    else if (t == "Dieletrics") { m.type = RT_MAT_DIELECTRIC; as_vec3(n.get("tint"), "tint", m.rgb); m.param = as_f64(need(n, "refraction_index"), "refraction_index"); }
    else if (t == "Emissive") { m.type = RT_MAT_EMISSIVE; as_vec3(n.get("color"), "color", m.rgb); }
    else if (t == "Texture") { m.type = RT_MAT_TEXTURE; m.texture_path = as_str(need(n, "texture_path"), "texture_path"); }
    else raise("unknown variant `%s`, expected one of `Lambertian`, `Metal`, `Dieletrics`, `Texture`, `Emissive` (line %d)", t.c_str(), n.line);
    return m;
}

struct TriangleH {           // mesh.rs:351-356
    uint64_t v[3];
    bool has_uv = false;
    uint64_t uv[3] = {0, 0, 0};
    int64_t material_idx = -1;   // usize::MAX in the reference before any usemtl
};
struct MeshH {               // mesh.rs:295-302
    std::vector<TriangleH> triangles;
    std::vector<MaterialH> materials;
    std::vector<double> positions;   // 3 per vertex
    std::vector<double> normals;     // loaded, never used
    std::vector<double> uvs;         // 2 per vt
};

// fast token scanner for OBJ lines (split_whitespace)
struct Tokens {
    const char* p;
    const char* e;
    bool next(const char*& b, const char*& en) {
        while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) p++;
        if (p >= e) return false;
        b = p;
        while (p < e && !(*p == ' ' || *p == '\t' || *p == '\r')) p++;
        en = p;
        return true;
    }
};
double tok_f64(Tokens& t, int line) {
    const char *b, *e;
    if (!t.next(b, e)) raise("obj line %d: missing number", line);
    std::string s(b, e);
    char* end = nullptr;
    double v = strtod(s.c_str(), &end);
    if (end != s.c_str() + s.size()) raise("obj line %d: '%s' is not a number", line, s.c_str());
    return v;
}
uint64_t parse_usize(const char* b, const char* e, int line) {
    if (b >= e) raise("obj line %d: missing index", line);
    uint64_t v = 0;
    const char* p = b;
    if (*p == '+') p++;
    if (p >= e) raise("obj line %d: bad index", line);
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') raise("obj line %d: '%s' is not an unsigned integer", line, std::string(b, e).c_str());
        v = v * 10 + (uint64_t)(*p - '0');
    }
    return v;
}

// mesh.rs:22-80 load_obj.  Panics of the reference become errors.
std::vector<MeshH> load_obj(const std::string& path, const std::map<std::string, MaterialH>& materials) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) raise("cannot open mesh '%s'", path.c_str());
    std::vector<char> data;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        data.resize(sz > 0 ? (size_t)sz : 0);
        if (sz > 0 && fread(data.data(), 1, (size_t)sz, f) != (size_t)sz) { fclose(f); raise("cannot read mesh '%s'", path.c_str()); }
        fclose(f);
    }
    std::vector<MeshH> result;
    uint64_t vcount = 0, ncount = 0, uvcount = 0;   // IndexHelper, mesh.rs:237-290
    int64_t material_index = -1;
    int line = 0;
    const char* p = data.data();
    const char* end = p + data.size();
    auto local = [&](uint64_t file_idx, uint64_t base, int ln) -> uint64_t {
        if (file_idx < base + 1) raise("obj line %d: index %llu underflows (mesh.rs:269-289)", ln, (unsigned long long)file_idx);
        return file_idx - base - 1;
    };
    while (p < end) {
        const char* le = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!le) le = end;
        ++line;
        Tokens t{p, le};
        p = le + 1;
        const char *b, *e;
        if (!t.next(b, e)) continue;
        size_t hl = (size_t)(e - b);
        auto head_is = [&](const char* s) { return hl == strlen(s) && !memcmp(b, s, hl); };
        if (head_is("o")) {                                       // mesh.rs:38-48
            if (!result.empty()) {
                vcount += result.back().positions.size() / 3;
                uvcount += result.back().uvs.size() / 2;
                ncount += result.back().normals.size() / 3;
            }
            result.emplace_back();
            material_index = -1;
        } else if (head_is("v") || head_is("vn")) {               // mesh.rs:49-58
            if (result.empty()) raise("obj line %d: vertex data before any 'o' line (the reference panics, mesh.rs:29,50)", line);
            bool is_n = hl == 2;
            double x = tok_f64(t, line), y = tok_f64(t, line), z = tok_f64(t, line);
            auto& dst = is_n ? result.back().normals : result.back().positions;
            dst.push_back(x); dst.push_back(y); dst.push_back(z);
        } else if (head_is("vt")) {                               // mesh.rs:59-63
            if (result.empty()) raise("obj line %d: vertex data before any 'o' line", line);
            double u = tok_f64(t, line), v = tok_f64(t, line);
            result.back().uvs.push_back(u); result.back().uvs.push_back(v);
        } else if (head_is("f")) {                                // mesh.rs:64-69, 110-183
            if (result.empty()) raise("obj line %d: face before any 'o' line", line);
            std::vector<std::pair<const char*, const char*>> toks;
            while (t.next(b, e)) toks.push_back({b, e});
            if (toks.empty()) raise("obj line %d: empty face", line);
            std::string first(toks[0].first, toks[0].second);
            TriangleH tr;
            tr.material_idx = material_index;
            if (first.find("//") != std::string::npos) {
                raise("obj line %d: 'v//vn' faces cannot be parsed by the reference (mesh.rs:206-210); unsupported", line);
            } else if (first.find('/') == std::string::npos) {    // VPos: exactly three tokens, extras dropped
                if (toks.size() < 3) raise("obj line %d: face needs three vertices", line);
                for (int k = 0; k < 3; ++k) tr.v[k] = local(parse_usize(toks[k].first, toks[k].second, line), vcount, line);
            } else if (first.find('/') == first.rfind('/')) {     // VPosUv: every token; a fourth panics
                if (toks.size() > 3) raise("obj line %d: 'v/vt' face with more than three vertices (the reference panics, mesh.rs:198-213)", line);
                if (toks.size() < 3) raise("obj line %d: face needs three vertices", line);
                for (int k = 0; k < 3; ++k) {
                    const char* s = (const char*)memchr(toks[k].first, '/', (size_t)(toks[k].second - toks[k].first));
                    if (!s) raise("obj line %d: expected v/vt", line);
                    tr.v[k] = local(parse_usize(toks[k].first, s, line), vcount, line);
                    tr.uv[k] = local(parse_usize(s + 1, toks[k].second, line), uvcount, line);
                }
                tr.has_uv = true;
            } else {                                              // VPosUvN: exactly three tokens
                if (toks.size() < 3) raise("obj line %d: face needs three vertices", line);
                for (int k = 0; k < 3; ++k) {
                    const char* s1 = (const char*)memchr(toks[k].first, '/', (size_t)(toks[k].second - toks[k].first));
                    const char* s2 = s1 ? (const char*)memchr(s1 + 1, '/', (size_t)(toks[k].second - s1 - 1)) : nullptr;
                    if (!s1 || !s2) raise("obj line %d: expected v/vt/vn", line);
                    tr.v[k] = local(parse_usize(toks[k].first, s1, line), vcount, line);
                    tr.uv[k] = local(parse_usize(s1 + 1, s2, line), uvcount, line);
                    (void)local(parse_usize(s2 + 1, toks[k].second, line), ncount, line);
                }
                tr.has_uv = true;
            }
            result.back().triangles.push_back(tr);
        } else if (head_is("usemtl")) {                           // mesh.rs:70-74
            if (result.empty()) raise("obj line %d: usemtl before any 'o' line", line);
            if (!t.next(b, e)) raise("obj line %d: usemtl without a name", line);
            std::string name(b, e);
            auto it = materials.find(name);
            if (it == materials.end()) raise("obj line %d: material '%s' is not listed in the scene file (mesh.rs:71)", line, name.c_str());
            result.back().materials.push_back(it->second);
            material_index = (int64_t)result.back().materials.size() - 1;
        }
    }
    return result;
}

void vsub(const double a[3], const double b[3], double o[3]) { for (int k = 0; k < 3; ++k) o[k] = a[k] - b[k]; }
double vdot(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
void vcross(const double a[3], const double b[3], double o[3]) {
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
void vnormalized(const double a[3], double o[3]) {   // vec3.rs:115-121
    double n = std::sqrt(vdot(a, a));
    for (int k = 0; k < 3; ++k) o[k] = n != 0.0 ? a[k] / n : a[k];
}

// camera.rs:26-54
void camera_new(const double eye[3], const double look_at[3], const double up[3], double fovy, uint32_t width, uint32_t height,
                double o_eye[3], double o_h[3], double o_v[3], double o_ll[3]) {
    double view_vec[3], view[3];
    vsub(look_at, eye, view_vec);
    double distance = std::sqrt(vdot(view_vec, view_vec));
    for (int k = 0; k < 3; ++k) view[k] = view_vec[k] / distance;
    double aspect = (double)width / (double)height;
    double image_height = 2.0 * distance * std::tan(0.5 * (fovy * (M_PI / 180.0)));
    double image_width = image_height * aspect;
    double c[3], n[3];
    vcross(view, up, c); vnormalized(c, n);
    for (int k = 0; k < 3; ++k) o_h[k] = n[k] * image_width / (double)width;
    vcross(o_h, view, c); vnormalized(c, n);
    for (int k = 0; k < 3; ++k) o_v[k] = n[k] * image_height / (double)height;
    for (int k = 0; k < 3; ++k) {
        o_ll[k] = look_at[k] - o_h[k] * (0.5 * (double)width) - o_v[k] * (0.5 * (double)height);
        o_eye[k] = eye[k];
    }
}

}  // namespace

// flattened arrays + the desc that points into them
struct rt_host_scene {
    rt_scene_desc desc{};
    std::vector<int> object_kinds;
    std::vector<int32_t> mat_type, mat_texture;
    std::vector<double> mat_rgb, mat_param;
    std::vector<uint32_t> tex_w, tex_h;
    std::vector<std::vector<uint8_t>> tex_data;
    std::vector<const uint8_t*> tex_ptr;
    std::vector<double> sph_c, sph_r;
    std::vector<int32_t> sph_m, sph_o;
    std::vector<double> pl_c, pl_n;
    std::vector<int32_t> pl_m, pl_o;
    std::vector<double> tri_v, tri_uv;
    std::vector<uint8_t> tri_has_uv;
    std::vector<int32_t> tri_m, tri_o, tri_i;
    std::vector<double> lights;

    int add_material(const MaterialH& m) {
        mat_type.push_back(m.type);
        for (int k = 0; k < 3; ++k) mat_rgb.push_back(m.rgb[k]);
        mat_param.push_back(m.param);
        mat_texture.push_back(m.texture);
        return (int)mat_type.size() - 1;
    }
    void finish() {
        desc.n_materials = (uint32_t)mat_type.size();
        desc.mat_type = mat_type.data(); desc.mat_rgb = mat_rgb.data(); desc.mat_param = mat_param.data(); desc.mat_texture = mat_texture.data();
        tex_ptr.clear();
        for (auto& t : tex_data) tex_ptr.push_back(t.data());
        desc.n_textures = (uint32_t)tex_w.size();
        desc.tex_width = tex_w.data(); desc.tex_height = tex_h.data(); desc.tex_rgb8 = tex_ptr.data();
        desc.n_spheres = (uint32_t)sph_r.size();
        desc.sphere_center = sph_c.data(); desc.sphere_radius = sph_r.data(); desc.sphere_material = sph_m.data(); desc.sphere_object = sph_o.data();
        desc.n_planes = (uint32_t)pl_m.size();
        desc.plane_center = pl_c.data(); desc.plane_normal = pl_n.data(); desc.plane_material = pl_m.data(); desc.plane_object = pl_o.data();
        desc.n_triangles = tri_m.size();
        desc.tri_vertices = tri_v.data();
        bool any_uv = false;
        for (uint8_t b : tri_has_uv) any_uv |= b != 0;
        desc.tri_uv = any_uv ? tri_uv.data() : nullptr;
        desc.tri_has_uv = any_uv ? tri_has_uv.data() : nullptr;
        desc.tri_material = tri_m.data(); desc.tri_object = tri_o.data(); desc.tri_index = tri_i.data();
        desc.n_lights = (uint32_t)(lights.size() / 3);
        desc.light_point = lights.data();
    }
};

namespace {

std::string join_path(const std::string& base, const std::string& rel) {
    namespace fs = std::filesystem;
    fs::path r(rel);
    if (r.is_absolute() || base.empty()) return r.string();
    return (fs::path(base) / r).lexically_normal().string();
}

std::unique_ptr<rt_host_scene> load_scene(const std::string& yaml_path, const std::string& base_dir) {
    std::ifstream in(yaml_path);
    if (!in) raise("cannot open scene '%s'", yaml_path.c_str());
    YamlParser yp(in);
    Node root = yp.parse();
    if (root.kind != Node::Map) raise("scene file: expected a mapping at the top level");
    auto hs = std::make_unique<rt_host_scene>();
    rt_scene_desc& d = hs->desc;
    // scene.rs:18-27
    const Node* cam = need(root, "camera");
    uint64_t w = as_usize(root.get("width"), "width"), h = as_usize(root.get("height"), "height");
    if (w == 0 || h == 0 || w > 65536 || h > 65536) raise("width/height out of range");
    d.width = (uint32_t)w; d.height = (uint32_t)h;
    as_vec3(root.get("background"), "background", d.background);
    double eye[3], look_at[3], up[3];
    if (cam->kind != Node::Map) raise("camera: expected a mapping");
    as_vec3(cam->get("eye"), "eye", eye); as_vec3(cam->get("look_at"), "look_at", look_at); as_vec3(cam->get("up"), "up", up);
    double fovy = as_f64(need(*cam, "fovy"), "fovy");
    camera_new(eye, look_at, up, fovy, d.width, d.height, d.cam_eye, d.cam_horizontal, d.cam_vertical, d.cam_lower_left);   // raytrace.rs:117-124

    const Node* objs = need(root, "objects");
    if (objs->kind != Node::Seq) raise("objects: expected a sequence");
    int32_t oi = 0;
    for (const Node& o : objs->seq) {
        if (o.kind != Node::Map) raise("object %d: expected a mapping", oi);
        std::string type = as_str(need(o, "type"), "object type");   // scene.rs:83-89
        if (type == "Sphere") {                                      // scene.rs:174-179
            double c[3];
            as_vec3(o.get("center"), "center", c);
            double r = as_f64(need(o, "radius"), "radius");
            MaterialH m = parse_material(*need(o, "material"));
            if (m.type == RT_MAT_TEXTURE) raise("object %d: a Texture material on a sphere has no uv (the reference panics, materials.rs:235)", oi);
            for (int k = 0; k < 3; ++k) hs->sph_c.push_back(c[k]);
            hs->sph_r.push_back(r); hs->sph_m.push_back(hs->add_material(m)); hs->sph_o.push_back(oi);
            if (m.type == RT_MAT_EMISSIVE) for (int k = 0; k < 3; ++k) hs->lights.push_back(c[k]);   // scene.rs:128
            hs->object_kinds.push_back(0);
        } else if (type == "Plane") {                                // scene.rs:181-186
            double c[3], n[3];
            as_vec3(o.get("center"), "center", c); as_vec3(o.get("normal"), "normal", n);
            MaterialH m = parse_material(*need(o, "material"));
            if (m.type == RT_MAT_TEXTURE) raise("object %d: a Texture material on a plane has no uv (the reference panics, materials.rs:235)", oi);
            for (int k = 0; k < 3; ++k) { hs->pl_c.push_back(c[k]); hs->pl_n.push_back(n[k]); }
            hs->pl_m.push_back(hs->add_material(m)); hs->pl_o.push_back(oi);
            if (m.type == RT_MAT_EMISSIVE) for (int k = 0; k < 3; ++k) hs->lights.push_back(c[k]);   // scene.rs:127
            hs->object_kinds.push_back(1);
        } else if (type == "Mesh") {                                 // scene.rs:158-172
            std::string path = join_path(base_dir, as_str(need(o, "path"), "path"));
            const Node* mats = need(o, "materials");
            if (mats->kind != Node::Map) raise("object %d: materials: expected a mapping", oi);
            std::map<std::string, MaterialH> mm;
            for (auto& kv : mats->map) mm[kv.first] = parse_material(kv.second);
            std::vector<MeshH> meshes = load_obj(path, mm);
            if (meshes.empty()) raise("object %d: '%s' holds no 'o' object (the reference panics, scene.rs:170)", oi, path.c_str());
            // Only the FIRST object of the file is kept (scene.rs:170).  Additive key `all_objects: true` (SURVEY
            // 8f-2; the reference's serde model ignores unknown keys): the file's other `o` objects are appended
            // to the first, so the whole file is one scene object with triangle ids in file order.
            if (const Node* all = o.get("all_objects")) {
                const std::string v = as_str(all, "all_objects");
                if (v == "true" || v == "True" || v == "yes") {
                    for (size_t mi = 1; mi < meshes.size(); ++mi) {
                        MeshH& src = meshes[mi];
                        const uint64_t v0 = meshes[0].positions.size() / 3, u0 = meshes[0].uvs.size() / 2;
                        const int64_t m0 = (int64_t)meshes[0].materials.size();
                        for (TriangleH t : src.triangles) {
                            for (int k = 0; k < 3; ++k) { t.v[k] += v0; if (t.has_uv) t.uv[k] += u0; }
                            if (t.material_idx >= 0) t.material_idx += m0;
                            meshes[0].triangles.push_back(t);
                        }
                        meshes[0].positions.insert(meshes[0].positions.end(), src.positions.begin(), src.positions.end());
                        meshes[0].uvs.insert(meshes[0].uvs.end(), src.uvs.begin(), src.uvs.end());
                        meshes[0].materials.insert(meshes[0].materials.end(), src.materials.begin(), src.materials.end());
                    }
                } else if (v != "false" && v != "False" && v != "no") {
                    raise("object %d: all_objects: expected true or false", oi);
                }
            }
            MeshH& mesh = meshes[0];
            // Scene::precompute, scene.rs:56-66: textures of mesh materials
            std::vector<int> global_mat;
            for (MaterialH& m : mesh.materials) {
                if (m.type == RT_MAT_TEXTURE) {
                    std::vector<uint8_t> px;
                    uint32_t tw = 0, th = 0;
                    read_png(join_path(base_dir, m.texture_path), px, tw, th);
                    hs->tex_w.push_back(tw); hs->tex_h.push_back(th); hs->tex_data.push_back(std::move(px));
                    m.texture = (int)hs->tex_w.size() - 1;
                }
                global_mat.push_back(hs->add_material(m));
            }
            const uint64_t nv = mesh.positions.size() / 3, nuv = mesh.uvs.size() / 2;
            bool light_done = false, is_light = false;
            for (auto& m : mesh.materials) is_light |= m.type == RT_MAT_EMISSIVE;     // scene.rs:102-109
            int32_t ti = 0;
            for (const TriangleH& t : mesh.triangles) {
                if (t.material_idx < 0) raise("object %d: triangle %d precedes any usemtl (the reference panics when it is hit, mesh.rs:31)", oi, ti);
                for (int k = 0; k < 3; ++k) {
                    if (t.v[k] >= nv) raise("object %d: triangle %d: vertex index out of range", oi, ti);
                    for (int c = 0; c < 3; ++c) hs->tri_v.push_back(mesh.positions[3 * t.v[k] + c]);
                }
                const MaterialH& tm = mesh.materials[(size_t)t.material_idx];
                if (t.has_uv) {
                    for (int k = 0; k < 3; ++k) {
                        if (t.uv[k] >= nuv) raise("object %d: triangle %d: uv index out of range", oi, ti);
                        hs->tri_uv.push_back(mesh.uvs[2 * t.uv[k]]); hs->tri_uv.push_back(mesh.uvs[2 * t.uv[k] + 1]);
                    }
                } else {
                    if (tm.type == RT_MAT_TEXTURE) raise("object %d: triangle %d has a Texture material but no vt (the reference panics, materials.rs:235)", oi, ti);
                    for (int k = 0; k < 6; ++k) hs->tri_uv.push_back(0.0);
                }
                hs->tri_has_uv.push_back(t.has_uv ? 1 : 0);
                hs->tri_m.push_back(global_mat[(size_t)t.material_idx]); hs->tri_o.push_back(oi); hs->tri_i.push_back(ti);
                if (is_light && !light_done && tm.type == RT_MAT_EMISSIVE) {          // scene.rs:129-143, sample_points[0]
                    const double* v = &hs->tri_v[hs->tri_v.size() - 9];
                    for (int c = 0; c < 3; ++c) hs->lights.push_back((v[c] + v[3 + c] + v[6 + c]) / 3.0);
                    light_done = true;
                }
                ti++;
            }
            if (is_light && !light_done) raise("object %d: mesh lists an Emissive material no triangle uses (the reference panics, raytrace.rs:81)", oi);
            hs->object_kinds.push_back(2);
        } else {
            raise("unknown variant `%s`, expected one of `Sphere`, `Plane`, `Mesh` (line %d)", type.c_str(), o.line);
        }
        oi++;
    }
    hs->finish();
    return hs;
}

int ssaa_parse(const std::string& s, int32_t* mode, uint32_t* n, std::string& err) {   // anti_aliasing.rs:50-83
    std::vector<std::string> parts;
    size_t pos = 0;
    for (;;) {
        size_t q = s.find(':', pos);
        if (q == std::string::npos) { parts.push_back(s.substr(pos)); break; }
        parts.push_back(s.substr(pos, q - pos));
        pos = q + 1;
    }
    auto parse_res = [&](const std::string& t, uint32_t* out) {
        size_t i = 0;
        if (i < t.size() && t[i] == '+') i++;
        if (i >= t.size()) return false;
        uint64_t v = 0;
        for (; i < t.size(); ++i) {
            if (t[i] < '0' || t[i] > '9') return false;
            v = v * 10 + (uint64_t)(t[i] - '0');
            if (v > 0xFFFFFFFFull) return false;
        }
        *out = (uint32_t)v;
        return true;
    };
    const std::string& m = parts[0];
    if (m == "jitter" || m == "uniform" || m == "grid") {
        if (parts.size() < 2) { err = "no arguments supplied, need resolution for " + m; return 1; }
        if (!parse_res(parts[1], n)) { err = "resolution has to be an integer"; return 1; }
        // anti_aliasing.rs:75: "uniform" also returns SuperSampling::Jitter
        *mode = (m == "grid") ? RT_SSAA_UNIFORM : RT_SSAA_JITTER;
        return 0;
    }
    err = "unknown method";
    return 1;
}

}  // namespace

extern "C" {

rt_host_scene* rt_host_scene_load(const char* yaml_path, const char* base_dir, char* err, size_t err_len) {
    try {
        if (!yaml_path) raise("yaml_path is NULL");
        return load_scene(yaml_path, base_dir ? base_dir : "").release();
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return nullptr;
    }
}
void rt_host_scene_free(rt_host_scene* s) { delete s; }
const rt_scene_desc* rt_host_scene_desc(const rt_host_scene* s) { return s ? &s->desc : nullptr; }
uint32_t rt_host_scene_num_objects(const rt_host_scene* s) { return s ? (uint32_t)s->object_kinds.size() : 0; }
int rt_host_scene_object_kind(const rt_host_scene* s, uint32_t i) { return (s && i < s->object_kinds.size()) ? s->object_kinds[i] : -1; }

int rt_host_ssaa_parse(const char* text, int32_t* mode, uint32_t* n, char* err, size_t err_len) {
    std::string e;
    if (!text || !mode || !n) { set_err(err, err_len, "NULL argument"); return 1; }
    int rc = ssaa_parse(text, mode, n, e);
    if (rc) set_err(err, err_len, e);
    return rc;
}

void rt_host_camera(const double eye[3], const double look_at[3], const double up[3], double fovy, uint32_t width, uint32_t height,
                    double out_eye[3], double out_horizontal[3], double out_vertical[3], double out_lower_left[3]) {
    camera_new(eye, look_at, up, fovy, width, height, out_eye, out_horizontal, out_vertical, out_lower_left);
}

int rt_host_write_png(const char* path, const uint8_t* rgb8, uint32_t width, uint32_t height, char* err, size_t err_len) {
    try {
        if (!path || !rgb8) raise("NULL argument");
        write_png(path, rgb8, width, height);
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return 1;
    }
}
int rt_host_write_png16(const char* path, const uint16_t* rgb16, uint32_t width, uint32_t height, char* err, size_t err_len) {
    try {
        if (!path || !rgb16) raise("NULL argument");
        const size_t nv = (size_t)width * height * 3;
        std::vector<uint8_t> be(nv * 2);
        for (size_t i = 0; i < nv; ++i) { be[2 * i] = (uint8_t)(rgb16[i] >> 8); be[2 * i + 1] = (uint8_t)(rgb16[i] & 0xFFu); }
        write_png(path, be.data(), width, height, 2);
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return 1;
    }
}
int rt_host_read_png(const char* path, uint8_t** rgb8, uint32_t* width, uint32_t* height, char* err, size_t err_len) {
    try {
        if (!path || !rgb8 || !width || !height) raise("NULL argument");
        std::vector<uint8_t> px;
        read_png(path, px, *width, *height);
        *rgb8 = (uint8_t*)malloc(px.size());
        if (!*rgb8) raise("out of memory");
        memcpy(*rgb8, px.data(), px.size());
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return 1;
    }
}
void rt_host_free(void* p) { free(p); }

// Builds the BVH over the given boxes and checks what the closest-hit kernel relies on: every
// primitive is in exactly one leaf, and every (decoded, quantised) child box contains all the boxes
// of its subtree.
int rt_host_bvh_validate(uint64_t n, const float* boxes, uint32_t* n_nodes, uint32_t* depth, uint32_t* max_stack, char* err, size_t err_len) {
    try {
        if (!boxes && n) raise("boxes is NULL");
        std::vector<rtamd::PrimBox> pb(n);
        for (uint64_t i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) { pb[i].mn[k] = boxes[6 * i + k]; pb[i].mx[k] = boxes[6 * i + 3 + k]; }
        rtamd::Bvh bvh;
        rtamd::build_bvh(pb, bvh);
        if (n_nodes) *n_nodes = (uint32_t)bvh.nodes.size();
        if (depth) *depth = bvh.max_depth;
        if (max_stack) *max_stack = bvh.max_stack;
        if (n == 0) return 0;
        rtamd::validate_bvh(bvh, pb);
        return 0;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
        return 1;
    }
}

int rt_host_compute_image(const char* ssaa, uint32_t depth, const char* scene_path, const char* output_path, const char* base_dir,
                          int device_id, uint64_t seed, char* err, size_t err_len) {
    rt_context* ctx = nullptr;
    rt_host_scene* hs = nullptr;
    int rc = 1;
    try {
        if (!ssaa || !scene_path || !output_path) raise("NULL argument");
        int32_t mode = 0;
        uint32_t n = 0;
        std::string e;
        if (ssaa_parse(ssaa, &mode, &n, e)) raise("error parsing ssaa argument: %s", e.c_str());   // anti_aliasing.rs:37
        if (depth > 255) raise("depth must fit the reference's u8 (main.rs:15-16)");
        hs = load_scene(scene_path, base_dir ? base_dir : "").release();
        if (rt_context_create(device_id, &ctx)) raise("%s", rt_last_error(nullptr));
        if (rt_scene_upload(ctx, &hs->desc)) raise("%s", rt_last_error(ctx));
        rt_render_params p{};
        p.ssaa_mode = mode; p.ssaa_n = n; p.depth = depth; p.seed = seed;
        std::vector<uint8_t> rgb((size_t)hs->desc.width * hs->desc.height * 3);
        if (rt_render_image(ctx, &p, nullptr, rgb.data())) raise("%s", rt_last_error(ctx));
        write_png(output_path, rgb.data(), hs->desc.width, hs->desc.height);   // raytrace.rs:155
        rc = 0;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
    }
    if (ctx) rt_context_destroy(ctx);
    delete hs;
    return rc;
}

// The same on several GPUs of one process: one context per device; the scene is flattened and its BVH built
// once (rt_scene_upload_devices); the samples (or the 8x8 tiles) are split between the contexts like the ranks
// of bench.py and the partial frames are summed and resolved ON the first GPU (rt_render_image_devices: peer
// copies + add kernel + rt_resolve, raytrace.rs:144-150 / image.rs:37-43), one copy to the host.
int rt_host_compute_image_devices(const char* ssaa, uint32_t depth, const char* scene_path, const char* output_path, const char* base_dir,
                                  const int* device_ids, uint32_t n_devices, int split_tiles, uint64_t seed, char* err, size_t err_len) {
    rt_host_scene* hs = nullptr;
    std::vector<rt_context*> ctxs;
    int rc = 1;
    try {
        if (!ssaa || !scene_path || !output_path || !device_ids) raise("NULL argument");
        if (n_devices == 0 || n_devices > 64) raise("n_devices must be in 1..64");
        int32_t mode = 0;
        uint32_t n = 0;
        std::string e;
        if (ssaa_parse(ssaa, &mode, &n, e)) raise("error parsing ssaa argument: %s", e.c_str());
        if (depth > 255) raise("depth must fit the reference's u8 (main.rs:15-16)");
        hs = load_scene(scene_path, base_dir ? base_dir : "").release();
        for (uint32_t k = 0; k < n_devices; ++k) {
            rt_context* c = nullptr;
            if (rt_context_create(device_ids[k], &c)) raise("device %d: %s", device_ids[k], rt_last_error(nullptr));
            ctxs.push_back(c);
        }
        if (rt_scene_upload_devices(ctxs.data(), n_devices, &hs->desc)) raise("%s", rt_last_error(ctxs[0]));
        rt_render_params p{};
        p.ssaa_mode = mode; p.ssaa_n = n; p.depth = depth; p.seed = seed;
        std::vector<uint8_t> rgb((size_t)hs->desc.width * hs->desc.height * 3);
        if (rt_render_image_devices(ctxs.data(), n_devices, &p, split_tiles, nullptr, rgb.data())) raise("%s", rt_last_error(ctxs[0]));
        write_png(output_path, rgb.data(), hs->desc.width, hs->desc.height);   // raytrace.rs:155
        rc = 0;
    } catch (const std::exception& e) {
        set_err(err, err_len, e.what());
    }
    for (rt_context* c : ctxs) rt_context_destroy(c);
    delete hs;
    return rc;
}

}  // extern "C"
