#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <string>
#include "twhost.h"
using namespace twhost;
static unsigned seed = 777; static unsigned rnd() { seed = seed * 1664525u + 1013904223u; return seed; }
// fuzz_decode.cpp — mutation fuzz of the host layer's PNG / JPEG / PGM decoders (load_gray) under AddressSanitizer +
// UndefinedBehaviorSanitizer (CPU build only: `make asan_fuzz`).  usage: fuzz_decode <iterations> <scratch file> <seed files...>
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const int iters = atoi(argv[1]);
    const char* scratch = argv[2];
    long ok = 0, bad = 0;
    for (int fi = 3; fi < argc; fi++) {
        const char* f = argv[fi];
        FILE* fp = fopen(f, "rb"); if (!fp) return 3; fseek(fp, 0, SEEK_END); long n = ftell(fp); fseek(fp, 0, SEEK_SET);
        std::vector<unsigned char> orig(n); if (fread(orig.data(), 1, n, fp) != (size_t)n) return 1; fclose(fp);
        for (int it = 0; it < iters; it++) {
            std::vector<unsigned char> b = orig;
            int m = 1 + rnd() % 8;
            for (int k = 0; k < m; k++) b[rnd() % b.size()] = (unsigned char)rnd();
            if (it % 5 == 0) b.resize(8 + rnd() % (b.size() - 8));
            FILE* o = fopen(scratch, "wb"); fwrite(b.data(), 1, b.size(), o); fclose(o);
            std::vector<uint8_t> img; int w = 0, h = 0;
            if (load_gray(scratch, img, w, h) && img.size() == (size_t)w * h) ok++; else bad++;
        }
    }
    printf("asan fuzz: decoded %ld rejected %ld\n", ok, bad);
    return 0;
}
