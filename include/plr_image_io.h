/* DDS reader / writer of the reference's asset pipeline: the on-disk format of baked SDF volumes.
 *
 * Replaces  bool loadDDSFile(const std::filesystem::path&, ImageDescription*, std::vector<uint8_t>*)   (Common/ImageIO.cpp:342-431)
 *      and  void writeDDSFile(const std::filesystem::path&, const ImageDescription&, const std::vector<uint8_t>&)   (ImageIO.cpp:448-571).
 * File layout: magic 0x20534444 ("DDS "), the 124-byte DDS_header, for fourCC "DX10" the 20-byte DDS_headerDX10, then the raw texels
 * (x fastest, then y, then z; all mips consecutively). The writer always emits a DX10 header (formats RGBA8 and R16_sFloat, as the
 * reference); the reader understands DX10/R16_FLOAT and the legacy fourCCs DXT1 (BC1), DXT5 (BC3), ATI2 (BC5), as the reference.
 * All functions return PLR_OK (0) or a negative code; plr_last_error() (plr.h) explains. No GPU is involved. */
#ifndef PLR_IMAGE_IO_H
#define PLR_IMAGE_IO_H
#include "plr.h"

#ifdef __cplusplus
extern "C" {
#endif

/* writeDDSFile. data_size must be a multiple of 4 (the reference copies whole dwords, ImageIO.cpp:557-569). */
int plr_write_dds_file(const char* path, const plr_image_desc* desc, const void* data, size_t data_size);

/* loadDDSFile. Fills out_desc (type from height/depth, mipCount = Manual with the file's mip count, usage Sampled). The texel
 * payload (file size minus headers) is returned in *out_data_size; it is copied to out_data when out_data is non-null and
 * capacity suffices (call once with out_data = NULL to size the buffer). */
int plr_load_dds_file(const char* path, plr_image_desc* out_desc, void* out_data, size_t capacity, size_t* out_data_size);

/* in-memory forms of the same (used by the file functions): serialise to / parse from a byte buffer */
int plr_encode_dds(const plr_image_desc* desc, const void* data, size_t data_size, void* out_file, size_t capacity, size_t* out_file_size);
int plr_decode_dds(const void* file, size_t file_size, plr_image_desc* out_desc, size_t* out_data_offset, size_t* out_data_size);

#ifdef __cplusplus
}
#endif
#endif
