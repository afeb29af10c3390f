// musica-standalone <raw_path> <out_path> [--size N] [--levels L] [--device D] [--debug-dir DIR] [--reference-order] [--clahe]
//
// Drop-in for the reference's `maverick-standalone.exe <raw_path>.raw <out_path>.bmp`
// (test/standalone/README.md:3, test/standalone/main.cpp:30-87), which is what the metamorphic
// harness spawns (test/metamorphic_test/script.py:200-214, check=True on the exit code):
//   * exactly two positional arguments, else "MAIN ERROR: wrong number of arguments", exit 1;
//   * input = 256-byte header + 3072*3072 little-endian uint16 (size checked), main.cpp:54-75;
//   * init(3072) -> execute -> saveOutImage(out), exit 0; any failure prints
//     "MAIN ERROR: <msg>" to stderr and exits 1 (main.cpp:7-11).
// Extensions that keep the two-argument form compatible: --size / --levels (also MUSICA_SIZE /
// MUSICA_LEVELS in the environment), --device, --debug-dir (the debugProcess() dumps that the
// reference writes in non-NDEBUG builds, main.cpp:81-83), --reference-order (MUSICA_FLAG_REFERENCE_ORDER: the shaders'
// literal 25-tap accumulation order) and --clahe (#define ENABLE_CLAHE of include/vk_processing.h:13).
// One process = one execute: the context is created without the launch-geometry autotune and without graph capture
// (both pay off only over many executes of one context; the reference's harness spawns a process per image).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <memory>
#include <string>
#include <vector>

#include "../../include/musica.h"

#define ASSERT_MSG(cond, msg)                        \
    if (!(cond)) {                                   \
        fprintf(stderr, "MAIN ERROR: %s\n", msg);    \
        exit(1);                                     \
    }

int main(int argc, char* argv[]) {
    uint32_t imageSize = 3072;  // main.cpp:31
    uint32_t levels = 0;        // ceil(log2 N), src/vk_processing.cpp:1989
    int device = 0;
    const char* debugDir = nullptr;
    uint32_t flags = MUSICA_FLAG_ONE_SHOT;   // one execute per process: no autotune, no graph capture, one stream
    musica_tunables tun;   // the reference's #defines / constants (include/vk_processing.h:16-17, 39-49) as options
    musica_tunables_default(&tun);
    if (const char* e = getenv("MUSICA_SIZE")) imageSize = (uint32_t)atoi(e);
    if (const char* e = getenv("MUSICA_LEVELS")) levels = (uint32_t)atoi(e);

    for (int i = 0; i < argc; i++) printf("%d = %s\n", i, argv[i]);  // main.cpp:33-35

    std::vector<const char*> pos;
    for (int i = 1; i < argc; i++) {
        if (!strcmp(argv[i], "--size") && i + 1 < argc) imageSize = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--levels") && i + 1 < argc) levels = (uint32_t)atoi(argv[++i]);
        else if (!strcmp(argv[i], "--device") && i + 1 < argc) device = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--debug-dir") && i + 1 < argc) debugDir = argv[++i];
        else if (!strcmp(argv[i], "--reference-order")) flags |= MUSICA_FLAG_REFERENCE_ORDER;
        else if (!strcmp(argv[i], "--clahe")) flags |= MUSICA_FLAG_CLAHE;
        else if (!strcmp(argv[i], "--linear-low-contrast")) tun.linear_low_contrast = 1;     // #define LINEAR_LOW_CONTRAST_LEVELS_REDUCTION
        else if (!strcmp(argv[i], "--linear-high-contrast")) tun.linear_high_contrast = 1;   // #define LINEAR_HIGH_CONTRAST_LEVELS_REDUCTION
        else pos.push_back(argv[i]);
    }
    ASSERT_MSG(pos.size() == 2, "wrong number of arguments");  // main.cpp:37

    const std::string rawFile = pos[0], outFile = pos[1];
    printf("raw file %s\n", rawFile.c_str());
    printf("out file %s\n", outFile.c_str());

    const auto t0 = std::chrono::high_resolution_clock::now();
    ASSERT_MSG(musica_device_count() > 0, "failed to initialize vk state");   // main.cpp:45-46 (VulkanState::init): first HIP call = runtime start-up
    const auto t0h = std::chrono::high_resolution_clock::now();
    musica_params p;
    memset(&p, 0, sizeof(p));
    p.image_size = imageSize;
    p.levels = levels;
    p.batch = 1;
    p.device = device;
    p.flags = flags;
    musica_ctx* ctx = musica_create_ex(&p, &tun);
    ASSERT_MSG(ctx != nullptr, "failed to initialize vk processing");  // main.cpp:51 (message kept)
    const auto t0c = std::chrono::high_resolution_clock::now();

    std::unique_ptr<uint16_t[]> pixels(new uint16_t[(size_t)imageSize * imageSize]);   // not zero-filled: the file fills it
    {
        FILE* f = fopen(rawFile.c_str(), "rb");
        ASSERT_MSG(f != nullptr, "failed to load file");  // main.cpp:55
        fclose(f);
    }
    ASSERT_MSG(musica_read_raw(rawFile.c_str(), imageSize, pixels.get()),
               "the image data don't match the actual image size");  // main.cpp:60
    const auto t1 = std::chrono::high_resolution_clock::now();

    ASSERT_MSG(musica_execute(ctx, pixels.get()), "processing failed");  // main.cpp:77
    const auto t2 = std::chrono::high_resolution_clock::now();

    ASSERT_MSG(musica_save_out_image(ctx, 0, outFile.c_str()), "failed to save out image");  // main.cpp:79
    if (debugDir) ASSERT_MSG(musica_debug_process(ctx, 0, debugDir), "failed to debug process");  // main.cpp:81-83
    const auto t3 = std::chrono::high_resolution_clock::now();

    auto ms = [](std::chrono::high_resolution_clock::time_point a, std::chrono::high_resolution_clock::time_point b) {
        return std::chrono::duration<float, std::chrono::milliseconds::period>(b - a).count();
    };
    // the reference prints one per-stage timing line per execute (src/vk_processing.cpp:2585-2595)
    printf("init: %.2f \t exec: %.2f \t save: %.2f \t tot: %.2f \n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t0, t3));
    // init = HIP runtime start-up + device allocation (create) and reading the raw file; exec = H2D + the pipeline + the final wait
    printf("hip start-up: %.2f \t create: %.2f \t read: %.2f \n", ms(t0, t0h), ms(t0h, t0c), ms(t0c, t1));

    musica_destroy(ctx);  // main.cpp:85-86
    printf("cleanup: %.2f \n", ms(t3, std::chrono::high_resolution_clock::now()));
    return 0;
}
