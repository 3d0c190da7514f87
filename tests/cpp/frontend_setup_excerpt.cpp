// Compile-only check of the drop-in boundary: the reference frontend's set-up sequence for descriptor set 0 - the calls
// RenderFrontend::setup makes before it creates any pass (RenderFrontend.cpp:280-311: setupGlobalShaderInfoLayout, setupGlobalShaderInfoResources) - written
// against include/plr_render_backend.hpp with the reference's own type and member names. If a member the frontend uses were missing from the shim this file would
// not compile; tests/test_foundations_cpu.py builds it with g++ (no GPU, no HIP headers needed: the shim only needs plr.h).
#include "../../include/plr_render_backend.hpp"
using namespace plrhost; // the shim keeps the reference's names in a namespace of its own (the forwarding header opens it)

// the frontend's binding numbers of set 0 (RenderFrontend.cpp:59-68; shader side: resources/shaders/global.inc)
const uint32_t globalUniformBufferBinding = 0;
const uint32_t globalSamplerAnisotropicRepeatBinding = 1;
const uint32_t globalSamplerNearestBlackBorderBinding = 2;
const uint32_t globalSamplerLinearRepeatBinding = 3;
const uint32_t globalSamplerLinearClampBinding = 4;
const uint32_t globalSamplerNearestClampBinding = 5;
const uint32_t globalSamplerLinearWhiteBorderBinding = 6;
const uint32_t globalSamplerNearestRepeatBinding = 7;
const uint32_t globalSamplerNearestWhiteBorderBinding = 8;
const uint32_t globalNoiseTextureBindingBinding = 9;

RenderBackend gRenderBackend; // RenderBackend.cpp:39

struct FrontendSetupExcerpt {
    UniformBufferHandle m_globalUniformBuffer;
    SamplerHandle m_sampler_anisotropicRepeat, m_sampler_nearestBlackBorder, m_sampler_linearRepeat, m_sampler_linearClamp, m_sampler_nearestClamp, m_sampler_linearWhiteBorder,
        m_sampler_nearestRepeat, m_sampler_nearestWhiteBorder;

    void setupGlobalShaderInfoLayout() {
        ShaderLayout globalLayout;
        globalLayout.uniformBufferBindings.push_back(globalUniformBufferBinding);
        globalLayout.sampledImageBindings.push_back(globalNoiseTextureBindingBinding);
        for (const uint32_t samplerBinding : {globalSamplerAnisotropicRepeatBinding, globalSamplerNearestBlackBorderBinding, globalSamplerLinearRepeatBinding, globalSamplerLinearClampBinding,
                                              globalSamplerNearestClampBinding, globalSamplerLinearWhiteBorderBinding, globalSamplerNearestRepeatBinding, globalSamplerNearestWhiteBorderBinding})
            globalLayout.samplerBindings.push_back(samplerBinding);
        gRenderBackend.setGlobalDescriptorSetLayout(globalLayout);
    }
    void setupGlobalShaderInfoResources() {
        RenderPassResources globalResources;
        globalResources.uniformBuffers = {UniformBufferResource(m_globalUniformBuffer, globalUniformBufferBinding)};
        globalResources.samplers = {SamplerResource(m_sampler_anisotropicRepeat, globalSamplerAnisotropicRepeatBinding), SamplerResource(m_sampler_nearestBlackBorder, globalSamplerNearestBlackBorderBinding),
                                    SamplerResource(m_sampler_linearRepeat, globalSamplerLinearRepeatBinding), SamplerResource(m_sampler_linearClamp, globalSamplerLinearClampBinding),
                                    SamplerResource(m_sampler_nearestClamp, globalSamplerNearestClampBinding), SamplerResource(m_sampler_linearWhiteBorder, globalSamplerLinearWhiteBorderBinding),
                                    SamplerResource(m_sampler_nearestRepeat, globalSamplerNearestRepeatBinding), SamplerResource(m_sampler_nearestWhiteBorder, globalSamplerNearestWhiteBorderBinding)};
        gRenderBackend.setGlobalDescriptorSetResources(globalResources);
    }
    void setup() {
        UniformBufferDescription globalBufferDesc;
        globalBufferDesc.size = 340;
        m_globalUniformBuffer = gRenderBackend.createUniformBuffer(globalBufferDesc);
        SamplerDescription sampler;
        sampler.interpolation = SamplerInterpolation::Linear;
        sampler.wrapping = SamplerWrapping::Repeat;
        m_sampler_linearRepeat = gRenderBackend.createSampler(sampler);
        setupGlobalShaderInfoLayout();   // "must be set once before creating renderpasses" (RenderBackend.h:72-73)
        setupGlobalShaderInfoResources();
        ComputePassDescription pass;
        pass.name = "Tonemap";
        pass.shaderDescription.srcPathRelative = "tonemapping.comp";
        (void)gRenderBackend.createComputePass(pass);
    }
};

int main() { return sizeof(FrontendSetupExcerpt) > 0 ? 0 : 1; } // never run: the test only compiles and links the symbols' declarations
