// Drop-in command line: the four flags of the reference's `Args`
// (/root/reference/src/main.rs:10-19) plus additive ones.
//   -s, --scene-path <PATH>   -o, --output-path <PATH>   -d, --depth <u8> [default 5]
//   --ssaa <method:N> [default uniform:2]
// additive: --device <id> [0], --seed <u64> [0], --base-dir <dir> (paths inside the scene
// file are resolved against it; default = working directory, like the reference)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/rtamd_host.h"

static void usage(FILE* f) {
    fprintf(f, "USAGE:\n    rtamd-render [OPTIONS] --scene-path <SCENE_PATH> --output-path <OUTPUT_PATH>\n\nOPTIONS:\n"
               "    -d, --depth <DEPTH>                [default: 5]\n    -o, --output-path <OUTPUT_PATH>\n"
               "    -s, --scene-path <SCENE_PATH>\n        --ssaa <SSAA>                  [default: uniform:2]\n"
               "        --device <ID>                  [default: 0]\n        --devices <ID,ID,...>          several GPUs, one context each\n"
               "        --split <spp|tiles>            how --devices share the frame [default: spp]\n        --seed <SEED>                  RNG seed; renders are repeatable (the reference seeds from the OS) [default: 0]\n"
               "        --base-dir <DIR>\n    -h, --help\n");
}

int main(int argc, char** argv) {
    std::string scene, output, ssaa = "uniform:2", base_dir;
    bool have_base = false;
    long depth = 5;
    int device = 0;
    std::vector<int> devices;      // --devices 0,1,2,...: one context per GPU
    bool split_tiles = false;      // --split spp (default) | tiles
    unsigned long long seed = 0;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i], v;
        auto value = [&](const char* name) -> bool {
            size_t eq = a.find('=');
            if (eq != std::string::npos) { v = a.substr(eq + 1); return true; }
            if (i + 1 >= argc) { fprintf(stderr, "error: The argument '%s' requires a value but none was supplied\n", name); exit(2); }
            v = argv[++i];
            return true;
        };
        auto is = [&](const char* s_, const char* l_) { return (s_ && a == s_) || a == l_ || a.rfind(std::string(l_) + "=", 0) == 0; };
        if (is("-s", "--scene-path")) { value("--scene-path"); scene = v; }
        else if (is("-o", "--output-path")) { value("--output-path"); output = v; }
        else if (is("-d", "--depth")) {
            value("--depth");
            char* end = nullptr;
            depth = strtol(v.c_str(), &end, 10);
            if (v.empty() || *end || depth < 0 || depth > 255) { fprintf(stderr, "error: Invalid value \"%s\" for '--depth <DEPTH>': not a u8\n", v.c_str()); return 2; }
        }
        else if (is(nullptr, "--ssaa")) { value("--ssaa"); ssaa = v; }
        else if (is(nullptr, "--device")) { value("--device"); device = atoi(v.c_str()); }
        else if (is(nullptr, "--devices")) {
            value("--devices");
            devices.clear();
            size_t pos = 0;
            while (pos <= v.size()) {
                size_t comma = v.find(',', pos);
                std::string tok = v.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
                char* end = nullptr;
                long d = strtol(tok.c_str(), &end, 10);
                if (tok.empty() || *end || d < 0) { fprintf(stderr, "error: Invalid value \"%s\" for '--devices <DEVICES>': expected a comma-separated list of GPU indices\n", v.c_str()); return 2; }
                devices.push_back((int)d);
                if (comma == std::string::npos) break;
                pos = comma + 1;
            }
        }
        else if (is(nullptr, "--split")) {
            value("--split");
            if (v == "tiles") split_tiles = true;
            else if (v == "spp") split_tiles = false;
            else { fprintf(stderr, "error: Invalid value \"%s\" for '--split <SPLIT>': expected spp or tiles\n", v.c_str()); return 2; }
        }
        else if (is(nullptr, "--seed")) { value("--seed"); seed = strtoull(v.c_str(), nullptr, 10); }
        else if (is(nullptr, "--base-dir")) { value("--base-dir"); base_dir = v; have_base = true; }
        else if (a == "-h" || a == "--help") { usage(stdout); return 0; }
        else { fprintf(stderr, "error: Found argument '%s' which wasn't expected\n", a.c_str()); usage(stderr); return 2; }
    }
    if (scene.empty() || output.empty()) {
        fprintf(stderr, "error: The following required arguments were not provided:\n%s%s\n", scene.empty() ? "    --scene-path <SCENE_PATH>\n" : "",
                output.empty() ? "    --output-path <OUTPUT_PATH>\n" : "");
        usage(stderr);
        return 2;
    }
    {   // clap validates --ssaa while parsing (main.rs:17-18)
        int32_t mode; uint32_t n; char e[256];
        if (rt_host_ssaa_parse(ssaa.c_str(), &mode, &n, e, sizeof(e))) {
            fprintf(stderr, "error: Invalid value \"%s\" for '--ssaa <SSAA>': error parsing ssaa argument: %s\n", ssaa.c_str(), e);
            return 2;
        }
    }
    char err[1024] = {0};
    if (!devices.empty()) {
        if (rt_host_compute_image_devices(ssaa.c_str(), (uint32_t)depth, scene.c_str(), output.c_str(), have_base ? base_dir.c_str() : nullptr,
                                          devices.data(), (uint32_t)devices.size(), split_tiles ? 1 : 0, seed, err, sizeof(err))) {
            fprintf(stderr, "error: %s\n", err);
            return 101;
        }
        return 0;
    }
    if (rt_host_compute_image(ssaa.c_str(), (uint32_t)depth, scene.c_str(), output.c_str(), have_base ? base_dir.c_str() : nullptr, device, seed, err, sizeof(err))) {
        fprintf(stderr, "error: %s\n", err);
        return 101;   // a Rust panic exits with 101
    }
    return 0;
}
