/*
 * ref_bmp.c — instantiates the REFERENCE's own BMP writer so tests can pin
 * the output-file byte layout against real reference code.
 *
 * The reference serialises its result with stbi_write_bmp(path, w, h, 1, data)
 * (src/vk_processing.cpp:2636-2642) from its vendored
 * dependencies/stb/stb_image_write.h.  That header is self-contained C, so it
 * is compiled here from where it lies under /root/reference (see Makefile,
 * target `ref`); no reference source is copied into this repository.
 * TEST INFRASTRUCTURE ONLY.
 */
#define STB_IMAGE_WRITE_IMPLEMENTATION
#include "stb_image_write.h"

int ref_write_bmp_gray(const char* path, int w, int h, const unsigned char* data) {
    return stbi_write_bmp(path, w, h, 1, data);
}

/* debugProcess writes its two plots with four components (src/vk_processing.cpp:2774-2780, :2799-2805). */
int ref_write_bmp_rgba(const char* path, int w, int h, const unsigned char* data) {
    return stbi_write_bmp(path, w, h, 4, data);
}
