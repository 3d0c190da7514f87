// DDS reader / writer (include/plr_image_io.h), the file format the reference's asset pipeline stores baked SDF volumes in
// (Common/ImageIO.cpp:118-147 header structs, :289-340 flag constants, :342-431 loadDDSFile, :433-571 writeDDSFile).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/plr_image_io.h"
#include "../backend.h"

namespace {

using plr::setLastError;

constexpr uint32_t kMagic = 0x20534444u; // "DDS "
// dwFlags, dwCaps, dwCaps2, pixel-format flags and fourCCs of the DDS specification (the subset the reference names, ImageIO.cpp:289-340)
constexpr uint32_t kFlagCaps = 0x1, kFlagHeight = 0x2, kFlagWidth = 0x4, kFlagPixelFormat = 0x1000, kFlagMipCount = 0x20000, kFlagDepth = 0x800000;
constexpr uint32_t kCapsComplex = 0x8, kCapsMipmap = 0x400000, kCapsTexture = 0x1000, kCaps2Volume = 0x200000;
constexpr uint32_t kPfFourCC = 0x4;
constexpr uint32_t kFourCCDXT1 = 0x31545844, kFourCCDXT5 = 0x35545844, kFourCCDX10 = 0x30315844, kFourCCBC5 = 0x32495441;
constexpr uint32_t kDxgiR16Float = 54, kDxgiRGBA8Unorm = 28;                 // DXGI_FORMAT enum positions
constexpr uint32_t kDimTexture1D = 2, kDimTexture2D = 3, kDimTexture3D = 4;  // D3D10_RESOURCE_DIMENSION

struct PixelFormat { uint32_t infoSize, flags, compressionCode, rgbBitCount, rMask, gMask, bMask, aMask; };
struct Header {
    uint32_t headerSize, flags, height, width, pitchOrLinearSize, depth, mipMapCount, reserved1[11];
    PixelFormat pixelFormat;
    uint32_t caps, caps2, caps3, caps4, reserved2;
};
struct HeaderDX10 { uint32_t dxgiFormat, resourceDimension, miscFlags, arraySize, miscFlags2; };
static_assert(sizeof(Header) == 124 && sizeof(HeaderDX10) == 20, "DDS header layout");

uint32_t mipCountFromResolution(uint32_t w, uint32_t h, uint32_t d) { // MathUtils.cpp:17-19
    return 1u + (uint32_t)std::floor(std::log2((double)std::max(std::max(w, h), d)));
}

int encode(const plr_image_desc* desc, const void* data, size_t dataSize, std::vector<uint8_t>& out) {
    if (!desc || (!data && dataSize)) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_encode_dds: null argument");
    if (dataSize % 4 != 0) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_encode_dds: data size must be a multiple of 4 bytes");
    Header h{};
    h.headerSize = sizeof(Header);
    h.flags = kFlagCaps | kFlagWidth | kFlagHeight | kFlagPixelFormat;
    if (desc->mip_count != PLR_MIP_ONE) h.flags |= kFlagMipCount;
    if (desc->depth != 1) h.flags |= kFlagDepth;
    h.height = desc->height; h.width = desc->width; h.pitchOrLinearSize = 0; h.depth = desc->depth;
    switch (desc->mip_count) {
        case PLR_MIP_ONE: h.mipMapCount = 1; break;
        case PLR_MIP_FULL_CHAIN: case PLR_MIP_FULL_CHAIN_ALREADY_IN_DATA: h.mipMapCount = mipCountFromResolution(desc->width, desc->height, desc->depth); break;
        case PLR_MIP_MANUAL: h.mipMapCount = desc->manual_mip_count; break;
        default: return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_encode_dds: unknown mip count mode");
    }
    h.pixelFormat.infoSize = sizeof(PixelFormat);
    h.pixelFormat.compressionCode = kFourCCDX10; // the legacy pixel format cannot name the formats needed; the DX10 header does (ImageIO.cpp:433-446)
    h.caps = kCapsTexture;
    if (h.mipMapCount != 1) h.caps |= kCapsMipmap | kCapsComplex;
    if (desc->depth != 1) { h.caps |= kCapsComplex; h.caps2 = kCaps2Volume; }
    HeaderDX10 x{};
    if (desc->format == PLR_FORMAT_RGBA8) x.dxgiFormat = kDxgiRGBA8Unorm;
    else if (desc->format == PLR_FORMAT_R16_SFLOAT) x.dxgiFormat = kDxgiR16Float;
    else return setLastError(PLR_ERR_UNSUPPORTED, "plr_encode_dds: only RGBA8 and R16_sFloat can be written (as in the reference)");
    x.arraySize = 1;
    x.resourceDimension = desc->depth == 1 ? (desc->height == 1 ? kDimTexture1D : kDimTexture2D) : kDimTexture3D;
    out.resize(4 + sizeof(Header) + sizeof(HeaderDX10) + dataSize);
    std::memcpy(out.data(), &kMagic, 4);
    std::memcpy(out.data() + 4, &h, sizeof(h));
    std::memcpy(out.data() + 4 + sizeof(h), &x, sizeof(x));
    if (dataSize) std::memcpy(out.data() + 4 + sizeof(h) + sizeof(x), data, dataSize);
    return PLR_OK;
}

int decode(const uint8_t* file, size_t fileSize, plr_image_desc* desc, size_t* dataOffset, size_t* dataSize) {
    if (!file || !desc || !dataOffset || !dataSize) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_decode_dds: null argument");
    if (fileSize < 4 + sizeof(Header)) return setLastError(PLR_ERR_INVALID_ARGUMENT, "DDS: file shorter than its header");
    uint32_t magic;
    std::memcpy(&magic, file, 4);
    if (magic != kMagic) return setLastError(PLR_ERR_INVALID_ARGUMENT, "DDS: bad magic number");
    Header h;
    std::memcpy(&h, file + 4, sizeof(h));
    std::memset(desc, 0, sizeof(*desc));
    desc->width = h.width; desc->height = h.height; desc->depth = std::max(h.depth, 1u);
    desc->type = desc->depth == 1 ? (desc->height == 1 ? PLR_IMAGE_1D : PLR_IMAGE_2D) : PLR_IMAGE_3D;
    desc->mip_count = PLR_MIP_MANUAL;
    desc->manual_mip_count = std::max(h.mipMapCount, 1u);
    desc->auto_create_mips = 0;
    desc->usage_flags = PLR_USAGE_SAMPLED;
    size_t offset = 4 + sizeof(Header);
    if (h.pixelFormat.compressionCode == kFourCCDX10) {
        if (fileSize < offset + sizeof(HeaderDX10)) return setLastError(PLR_ERR_INVALID_ARGUMENT, "DDS: file shorter than its DX10 header");
        HeaderDX10 x;
        std::memcpy(&x, file + offset, sizeof(x));
        offset += sizeof(x);
        if (x.dxgiFormat == kDxgiR16Float) desc->format = PLR_FORMAT_R16_SFLOAT;
        else return setLastError(PLR_ERR_UNSUPPORTED, "DDS unsupported texture format (DX10 header: only R16_FLOAT is read, as in the reference)");
    } else if (h.pixelFormat.compressionCode == kFourCCDXT1) desc->format = PLR_FORMAT_BC1;
    else if (h.pixelFormat.compressionCode == kFourCCDXT5) desc->format = PLR_FORMAT_BC3;
    else if (h.pixelFormat.compressionCode == kFourCCBC5) desc->format = PLR_FORMAT_BC5;
    else return setLastError(PLR_ERR_UNSUPPORTED, "DDS unsupported texture format");
    *dataOffset = offset;
    *dataSize = fileSize - offset; // data size is the file size without magic number and headers (:419-423)
    return PLR_OK;
}

} // namespace

extern "C" int plr_encode_dds(const plr_image_desc* desc, const void* data, size_t dataSize, void* outFile, size_t capacity, size_t* outFileSize) {
    std::vector<uint8_t> bytes;
    if (int rc = encode(desc, data, dataSize, bytes)) return rc;
    if (outFileSize) *outFileSize = bytes.size();
    if (outFile) {
        if (capacity < bytes.size()) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_encode_dds: output buffer too small");
        std::memcpy(outFile, bytes.data(), bytes.size());
    }
    return PLR_OK;
}

extern "C" int plr_decode_dds(const void* file, size_t fileSize, plr_image_desc* desc, size_t* dataOffset, size_t* dataSize) {
    return decode((const uint8_t*)file, fileSize, desc, dataOffset, dataSize);
}

extern "C" int plr_write_dds_file(const char* path, const plr_image_desc* desc, const void* data, size_t dataSize) {
    if (!path) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_write_dds_file: null path");
    std::vector<uint8_t> bytes;
    if (int rc = encode(desc, data, dataSize, bytes)) return rc;
    FILE* f = std::fopen(path, "wb");
    if (!f) return setLastError(PLR_ERR_INVALID_ARGUMENT, std::string("failed to open for writing: ") + path);
    const size_t n = std::fwrite(bytes.data(), 1, bytes.size(), f);
    const int closed = std::fclose(f);
    if (n != bytes.size() || closed != 0) return setLastError(PLR_ERR_INVALID_ARGUMENT, std::string("short write: ") + path);
    return PLR_OK;
}

extern "C" int plr_load_dds_file(const char* path, plr_image_desc* desc, void* outData, size_t capacity, size_t* outDataSize) {
    if (!path || !desc || !outDataSize) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_load_dds_file: null argument");
    FILE* f = std::fopen(path, "rb");
    if (!f) return setLastError(PLR_ERR_INVALID_ARGUMENT, std::string("failed to open image: ") + path);
    std::vector<uint8_t> bytes;
    uint8_t chunk[1 << 16];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof(chunk), f)) > 0) bytes.insert(bytes.end(), chunk, chunk + n);
    std::fclose(f);
    size_t offset = 0, size = 0;
    if (int rc = decode(bytes.data(), bytes.size(), desc, &offset, &size)) return rc;
    *outDataSize = size;
    if (outData) {
        if (capacity < size) return setLastError(PLR_ERR_INVALID_ARGUMENT, "plr_load_dds_file: output buffer too small");
        std::memcpy(outData, bytes.data() + offset, size);
    }
    return PLR_OK;
}
