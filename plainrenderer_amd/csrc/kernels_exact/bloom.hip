// Bloom chain for gfx950: bloomDownsample.comp, bloomUpsample.comp, applyBloom.comp
// (resources/shaders/, host Techniques/Bloom.cpp:56-143). R11G11B10 mips, 13-tap down / 9+4-tap up / lerp apply.
//
// Each lane produces one output texel; a wave covers a 64-texel row segment so the packed 4-byte stores and the
// (L1/L2-resident, quarter-size) source gathers stay coalesced. Bilinear taps are evaluated with the 8-bit sub-texel
// weights of the sampler contract; taps that land on texel centres or corners degenerate to 1 or 2 fetches, which is
// what the even-size mips of the chain produce for most taps (a fetch with an exactly zero weight is skipped: inputs
// are finite by contract, so t*0 contributes exactly 0).
#include "../backend.h"
#include "../device/shading_common.h"

namespace plr {

// linear-clamp sample of an R11G11B10 image, skipping zero-weight texels
PLR_DI vec3 bloomTap(const ImgView& im, float u, float v) {
    int i0, j0; float a, b;
    linearCoord(u * (float)im.w, &i0, &a);
    linearCoord(v * (float)im.h, &j0, &b);
    const uint32_t* base = (const uint32_t*)im.ptr;
    const int x0 = clampi(i0, im.w), x1 = clampi(i0 + 1, im.w);
    const size_t r0 = (size_t)clampi(j0, im.h) * (size_t)im.w, r1 = (size_t)clampi(j0 + 1, im.h) * (size_t)im.w;
    const float w00 = (1.f - a) * (1.f - b), w10 = a * (1.f - b), w01 = (1.f - a) * b, w11 = a * b;
    // same accumulation order as the full four-texel form: t00*w00 + t10*w10 + t01*w01 + t11*w11
    vec3 r = unpackR11G11B10(base[r0 + x0]) * w00; // w00 > 0 always (a, b < 1)
    if (w10 != 0.f) r = r + unpackR11G11B10(base[r0 + x1]) * w10;
    if (w01 != 0.f) r = r + unpackR11G11B10(base[r1 + x0]) * w01;
    if (w11 != 0.f) r = r + unpackR11G11B10(base[r1 + x1]) * w11;
    return r;
}

// bloomDownsample.comp:12-49
__global__ __launch_bounds__(256) void bloomDownsampleKernel(ImgView source, ImgView target, int coverW, int coverH, int yBase, int xBase) {
    const int x = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u)); // columns [xBase, coverW) (tile rendering: PassCtx::colSpan)
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= coverW || y >= coverH) return;
    const float uvx = ((float)x + 0.5f) / (float)target.w, uvy = ((float)y + 0.5f) / (float)target.h;
    const float tsx = 1.f / (float)source.w, tsy = 1.f / (float)source.h;
    vec3 color(0.f);
    color += bloomTap(source, uvx, uvy) * 0.125f;
    color += bloomTap(source, uvx + tsx * 0.5f, uvy + tsy * 0.5f) * 0.125f;
    color += bloomTap(source, uvx + tsx * 0.5f, uvy + tsy * -0.5f) * 0.125f;
    color += bloomTap(source, uvx + tsx * -0.5f, uvy + tsy * 0.5f) * 0.125f;
    color += bloomTap(source, uvx + tsx * -0.5f, uvy + tsy * -0.5f) * 0.125f;
    color += bloomTap(source, uvx + tsx * 1.5f, uvy + tsy * 0.f) * 0.0625f;
    color += bloomTap(source, uvx + tsx * -1.5f, uvy + tsy * 0.f) * 0.0625f;
    color += bloomTap(source, uvx + tsx * 0.f, uvy + tsy * 1.5f) * 0.0625f;
    color += bloomTap(source, uvx + tsx * 0.f, uvy + tsy * -1.5f) * 0.0625f;
    color += bloomTap(source, uvx + tsx * 1.5f, uvy + tsy * 1.5f) * 0.03125f;
    color += bloomTap(source, uvx + tsx * 1.5f, uvy + tsy * -1.5f) * 0.03125f;
    color += bloomTap(source, uvx + tsx * -1.5f, uvy + tsy * 1.5f) * 0.03125f;
    color += bloomTap(source, uvx + tsx * -1.5f, uvy + tsy * -1.5f) * 0.03125f;
    ((uint32_t*)target.ptr)[(size_t)y * (size_t)target.w + x] = packR11G11B10(color);
}

// columns [x0, w) and rows [y0, h) of the target covered by the recorded dispatch
static int coverage(const PassCtx& c, const ImgView& target, int* w, int* h, int* y0, int* x0) {
    const PassCtx::ColSpan cs = c.colSpan(target.w);
    *w = cs.x1; *x0 = cs.x0;
    const PassCtx::RowSpan rs = c.rowSpan(target.h);
    *h = rs.y1; *y0 = rs.y0;
    return (*w > *x0 && *h > *y0) ? 1 : 0;
}

static int launchBloomDownsample(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R11G11B10, "bloomDownsample target")) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "bloomDownsample source")) return rc;
    int w, h, y0, x0;
    if (!coverage(c, c.storage[0], &w, &h, &y0, &x0)) return 0;
    bloomDownsampleKernel<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.sampled[1], c.storage[0], w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("bloomDownsample.comp", launchBloomDownsample);

// bloomUpsample.comp:19-57
template <bool LOWEST>
__global__ __launch_bounds__(256) void bloomUpsampleKernel(ImgView source, ImgView previous, ImgView target, float blurRadius, int coverW, int coverH, int yBase, int xBase) {
    const int x = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= coverW || y >= coverH) return;
    const float tsx = 1.f / (float)source.w, tsy = 1.f / (float)source.h;
    const float sx = blurRadius * tsx, sy = blurRadius * tsy;
    const float uvx = ((float)x + 0.5f) / (float)target.w, uvy = ((float)y + 0.5f) / (float)target.h;
    vec3 color(0.f);
    color += bloomTap(source, uvx, uvy) * 0.25f;
    color += bloomTap(source, uvx + sx * 1.f, uvy + sy * 0.f) * 0.125f;
    color += bloomTap(source, uvx + sx * -1.f, uvy + sy * 0.f) * 0.125f;
    color += bloomTap(source, uvx + sx * 0.f, uvy + sy * 1.f) * 0.125f;
    color += bloomTap(source, uvx + sx * 0.f, uvy + sy * -1.f) * 0.125f;
    color += bloomTap(source, uvx + sx * 1.f, uvy + sy * 1.f) * 0.0625f;
    color += bloomTap(source, uvx + sx * 1.f, uvy + sy * -1.f) * 0.0625f;
    color += bloomTap(source, uvx + sx * -1.f, uvy + sy * 1.f) * 0.0625f;
    color += bloomTap(source, uvx + sx * -1.f, uvy + sy * -1.f) * 0.0625f;
    if (!LOWEST) {
        color += bloomTap(previous, uvx + tsx * 0.5f, uvy + tsy * 0.5f) * 0.25f;
        color += bloomTap(previous, uvx + tsx * 0.5f, uvy + tsy * -0.5f) * 0.25f;
        color += bloomTap(previous, uvx + tsx * -0.5f, uvy + tsy * 0.5f) * 0.25f;
        color += bloomTap(previous, uvx + tsx * -0.5f, uvy + tsy * -0.5f) * 0.25f;
    }
    ((uint32_t*)target.ptr)[(size_t)y * (size_t)target.w + x] = packR11G11B10(color);
}

static int launchBloomUpsample(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R11G11B10, "bloomUpsample target")) return rc;
    if (int rc = c.needSampled(2, F_R11G11B10, "bloomUpsample source")) return rc;
    const bool lowest = c.specBool(0, false);
    if (!lowest) if (int rc = c.needSampled(1, F_R11G11B10, "bloomUpsample targetPreviousMip")) return rc;
    if (c.push.size() < 4) return c.fail(-1, "bloomUpsample: push constant blurRadius missing");
    float blurRadius;
    std::memcpy(&blurRadius, c.push.data(), 4);
    int w, h, y0, x0;
    if (!coverage(c, c.storage[0], &w, &h, &y0, &x0)) return 0;
    const dim3 grid(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u));
    if (lowest) bloomUpsampleKernel<true><<<grid, 256, 0, c.stream>>>(c.sampled[2], c.sampled[2], c.storage[0], blurRadius, w, h, y0, x0);
    else bloomUpsampleKernel<false><<<grid, 256, 0, c.stream>>>(c.sampled[2], c.sampled[1], c.storage[0], blurRadius, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("bloomUpsample.comp", launchBloomUpsample);

// applyBloom.comp:16-30: target = mix(scene, bloom, strength), in place. The bloom image has the target's size, so the
// bilinear tap sits on a texel centre; it is still evaluated through the sampler path for exactness.
__global__ __launch_bounds__(256) void applyBloomKernel(ImgView target, ImgView bloom, float bloomStrength, int coverW, int coverH, int yBase, int xBase) {
    const int x = xBase + (int)(blockIdx.x * 64u + (threadIdx.x & 63u));
    const int y = yBase + (int)(blockIdx.y * 4u + (threadIdx.x >> 6));
    if (x >= coverW || y >= coverH) return;
    const float uvx = ((float)x + 0.5f) / (float)target.w, uvy = ((float)y + 0.5f) / (float)target.h;
    const vec3 b = bloomTap(bloom, uvx, uvy);
    uint32_t* px = (uint32_t*)target.ptr + (size_t)y * (size_t)target.w + x;
    const vec3 scene = unpackR11G11B10(*px);
    *px = packR11G11B10(vmix(scene, b, bloomStrength));
}

static int launchApplyBloom(const PassCtx& c) {
    if (int rc = c.needStorage(0, F_R11G11B10, "applyBloom target")) return rc;
    if (int rc = c.needSampled(1, F_R11G11B10, "applyBloom bloomTexture")) return rc;
    if (c.push.size() < 4) return c.fail(-1, "applyBloom: push constant bloomStrength missing");
    float strength;
    std::memcpy(&strength, c.push.data(), 4);
    int w, h, y0, x0;
    if (!coverage(c, c.storage[0], &w, &h, &y0, &x0)) return 0;
    applyBloomKernel<<<dim3(divUp((unsigned)(w - x0), 64u), divUp((unsigned)(h - y0), 4u)), 256, 0, c.stream>>>(c.storage[0], c.sampled[1], strength, w, h, y0, x0);
    PLR_CHECK_LAUNCH(c);
    return 0;
}
PLR_REGISTER_SHADER("applyBloom.comp", launchApplyBloom);

} // namespace plr
