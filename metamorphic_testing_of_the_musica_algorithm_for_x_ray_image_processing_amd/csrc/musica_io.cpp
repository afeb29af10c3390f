// musica_io.cpp — the two file formats of the reference's CLI contract (host only, no GPU):
//   input : test/standalone/main.cpp:54-75 — 256-byte header (ignored) + N*N little-endian uint16,
//           file size must equal 256 + 2*N*N exactly;
//   output: stbi_write_bmp(path, w, h, comp = 1, data) as called by saveOutImage
//           (src/vk_processing.cpp:2636-2642): 24-bpp BI_RGB, gray replicated to B, G, R, rows stored
//           bottom-up and padded to 4 bytes (dependencies/stb/stb_image_write.h:492-500, :451-476).
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/musica.h"

extern "C" int musica_read_raw(const char* path, uint32_t image_size, uint16_t* dst) {
    if (!path || !dst) return 0;
    FILE* f = fopen(path, "rb");
    if (!f) return 0;
    const long offset = 256;
    const long expected = offset + (long)image_size * image_size * 2;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    if (size != expected) {  // "the image data don't match the actual image size", main.cpp:57-60
        fclose(f);
        return 0;
    }
    fseek(f, offset, SEEK_SET);
    const size_t n = (size_t)image_size * image_size;
    // main.cpp:71-72 composes pixel i as bytes[2i + 1] << 8 | bytes[2i]: the file is little-endian, and so is every host this library
    // runs on (x86-64 beside an MI355X) — the bytes are read straight into the pixels; a big-endian build swaps them afterwards
    const size_t got = fread(dst, 1, n * 2, f);
    fclose(f);
    if (got != n * 2) return 0;
#if defined(__BYTE_ORDER__) && __BYTE_ORDER__ == __ORDER_BIG_ENDIAN__
    for (size_t i = 0; i < n; i++) dst[i] = (uint16_t)((dst[i] << 8) | (dst[i] >> 8));
#endif
    return 1;
}

static void le16(uint8_t* p, uint32_t v) { p[0] = (uint8_t)(v & 0xFF); p[1] = (uint8_t)((v >> 8) & 0xFF); }
static void le32(uint8_t* p, uint32_t v) { le16(p, v & 0xFFFF); le16(p + 2, v >> 16); }

// The 54 header bytes of stbi_write_bmp(path, w, h, comp = 1, data); returns the padded row size.
extern "C" uint32_t musica_bmp24_header(uint32_t w, uint32_t h, uint8_t hdr[54]) {
    const uint32_t pad = (uint32_t)(-(int32_t)(w * 3)) & 3u;
    const uint32_t row_bytes = w * 3 + pad;
    for (int i = 0; i < 54; i++) hdr[i] = 0;
    hdr[0] = 'B'; hdr[1] = 'M';
    le32(hdr + 2, 14 + 40 + row_bytes * h);  // file size
    le32(hdr + 10, 14 + 40);                 // pixel data offset
    le32(hdr + 14, 40);                      // BITMAPINFOHEADER
    le32(hdr + 18, w);
    le32(hdr + 22, h);
    le16(hdr + 26, 1);                       // planes
    le16(hdr + 28, 24);                      // bits per pixel; compression and the rest stay 0
    return row_bytes;
}
// A whole file image (header + pixel array as the device wrote it) in one write.
extern "C" int musica_write_file(const char* path, const uint8_t* bytes, size_t count) {
    if (!path || !bytes) return 0;
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    setvbuf(f, nullptr, _IONBF, 0);
    bool ok = fwrite(bytes, 1, count, f) == count;
    ok = (fclose(f) == 0) && ok;
    return ok ? 1 : 0;
}

extern "C" int musica_write_bmp_gray(const char* path, uint32_t w, uint32_t h, const uint8_t* data) {
    if (!path || !data) return 0;
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    setvbuf(f, nullptr, _IOFBF, 1 << 20);   // ~28 MB for a 3052^2 image: fewer, larger writes
    uint8_t hdr[54];
    const uint32_t row_bytes = musica_bmp24_header(w, h, hdr);
    bool ok = fwrite(hdr, 1, sizeof(hdr), f) == sizeof(hdr);
    std::vector<uint8_t> row(row_bytes, 0);
    for (int64_t j = (int64_t)h - 1; j >= 0 && ok; j--) {
        const uint8_t* src = data + (size_t)j * w;
        for (uint32_t i = 0; i < w; i++) row[3 * i] = row[3 * i + 1] = row[3 * i + 2] = src[i];
        ok = fwrite(row.data(), 1, row_bytes, f) == row_bytes;
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? 1 : 0;
}

// stbi_write_bmp(path, w, h, comp = 4, data), stb_image_write.h:501-509: BITMAPV4HEADER with BI_BITFIELDS and the masks
// R 0x00ff0000, G 0x0000ff00, B 0x000000ff, A 0xff000000; rows bottom-up, a pixel as B, G, R, A; no padding.
extern "C" int musica_write_bmp_rgba(const char* path, uint32_t w, uint32_t h, const uint8_t* data) {
    if (!path || !data) return 0;
    FILE* f = fopen(path, "wb");
    if (!f) return 0;
    uint8_t hdr[14 + 108] = {0};
    hdr[0] = 'B'; hdr[1] = 'M';
    le32(hdr + 2, 14 + 108 + w * h * 4);     // file size
    le32(hdr + 10, 14 + 108);                // pixel data offset
    le32(hdr + 14, 108);                     // BITMAPV4HEADER
    le32(hdr + 18, w);
    le32(hdr + 22, h);
    le16(hdr + 26, 1);                       // planes
    le16(hdr + 28, 32);                      // bits per pixel
    le32(hdr + 30, 3);                       // BI_BITFIELDS; the five words after it stay 0
    le32(hdr + 54, 0x00ff0000u);
    le32(hdr + 58, 0x0000ff00u);
    le32(hdr + 62, 0x000000ffu);
    le32(hdr + 66, 0xff000000u);             // colour space, endpoints and gammas stay 0
    bool ok = fwrite(hdr, 1, sizeof(hdr), f) == sizeof(hdr);
    std::vector<uint8_t> row((size_t)w * 4);
    for (int64_t j = (int64_t)h - 1; j >= 0 && ok; j--) {
        const uint8_t* src = data + (size_t)j * w * 4;
        for (uint32_t i = 0; i < w; i++) {
            row[4 * i] = src[4 * i + 2]; row[4 * i + 1] = src[4 * i + 1]; row[4 * i + 2] = src[4 * i]; row[4 * i + 3] = src[4 * i + 3];
        }
        ok = fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    ok = (fclose(f) == 0) && ok;
    return ok ? 1 : 0;
}
