// rt_post.hip -- what renderImage and Screen do to a finished frame: the bloom
// post-process (src/extra.cpp:214-316, called from src/render.cpp:48-50) and the
// conversion to 8-bit pixels in the byte order of the bitmap file
// (src/screen.cpp:44-55 through stb's stbi_write_bmp).  Both are one lane per
// pixel, bound by HBM traffic: bloom reads 12 B + k*k neighbours (L2-served) and
// writes 12 B per pixel, the packing reads 12 B and writes 4 B.
//
// The frame is in Screen order: pixel (x, y) of the reference's bottom-left
// based coordinates sits at flat index (H - 1 - y) * W + x (Screen::indexAt).

#include "rt_post.hpp"

namespace rt {

// computeHighpassFilter, src/extra.cpp:214-226: pixels darker than the threshold become black.
__global__ __launch_bounds__(256) void bloomHighpassKernel(const float* __restrict__ frame, float* __restrict__ bright, size_t nPixels,
    float threshold)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nPixels)
        return;
    const float r = frame[3 * i + 0], g = frame[3 * i + 1], b = frame[3 * i + 2];
    const float luminance = 0.2126f * r + 0.7152f * g + 0.0722f * b; // glm::dot: (x + y) + z
    const bool keep = !(luminance < threshold);
    bright[3 * i + 0] = keep ? r : 0.0f;
    bright[3 * i + 1] = keep ? g : 0.0f;
    bright[3 * i + 2] = keep ? b : 0.0f;
}

// The filter loop and the final sum of src/extra.cpp:296-315.  The reference filters
// in place, but for k >= 3 it only ever reads pixels it has not written yet (its
// window reaches forward in x and y only, and the one column that wraps into the
// next memory row lands on x = 0, which lies outside the filtered range), so reading
// the thresholded copy gives the same values.  Its window also reaches one row past
// the top of the image (flat index below zero, undefined behaviour there); such a
// read counts as black here, any other flat index inside the buffer is read as the
// reference's index arithmetic has it, wrap included.
__global__ __launch_bounds__(256) void bloomApplyKernel(float* __restrict__ frame, const float* __restrict__ bright, int W, int H, int k,
    BloomWeights w)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t nPixels = (size_t)W * H;
    if (i >= nPixels)
        return;
    const int x = (int)(i % (size_t)W), y = H - 1 - (int)(i / (size_t)W);
    float cr = bright[3 * i + 0], cg = bright[3 * i + 1], cb = bright[3 * i + 2];
    if (x >= k - 2 && x < W - k + 2 && y >= k - 2 && y < H - k + 2) {
        cr = cg = cb = 0.0f;
        for (int a = 0; a < k; a++) {
            for (int b = 0; b < k; b++) {
                const long long idx = (long long)(H - 1 - (y + b)) * W + (x + a);
                float pr = 0.0f, pg = 0.0f, pb = 0.0f;
                if (idx >= 0 && idx < (long long)nPixels)
                    pr = bright[3 * idx + 0], pg = bright[3 * idx + 1], pb = bright[3 * idx + 2];
                const float f = w.w[b]; // filter[a][b] only depends on b, see bloomWeights()
                cr = cr + f * pr;
                cg = cg + f * pg;
                cb = cb + f * pb;
            }
        }
    }
    frame[3 * i + 0] = frame[3 * i + 0] + cr;
    frame[3 * i + 1] = frame[3 * i + 1] + cg;
    frame[3 * i + 2] = frame[3 * i + 2] + cb;
}

// Screen::writeBitmapToFile's clamp and 8-bit conversion (glm::u8vec4(vec4(clamped, 1) * 255):
// truncation), laid out as stbi_write_bmp writes a 4-component image: rows bottom-up,
// bytes B, G, R, A.
__global__ __launch_bounds__(256) void packBmpKernel(const float* __restrict__ frame, uint8_t* __restrict__ out, int W, int H)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)W * H)
        return;
    const size_t row = i / (size_t)W, col = i % (size_t)W; // row 0 = top of the image
    const float* p = frame + 3 * i;
    const float r = fminf(fmaxf(p[0], 0.0f), 1.0f) * 255.0f, g = fminf(fmaxf(p[1], 0.0f), 1.0f) * 255.0f,
                b = fminf(fmaxf(p[2], 0.0f), 1.0f) * 255.0f;
    uchar4 px;
    px.x = (uint8_t)b, px.y = (uint8_t)g, px.z = (uint8_t)r, px.w = 255;
    reinterpret_cast<uchar4*>(out)[((size_t)H - 1 - row) * W + col] = px;
}

hipError_t launchBloom(hipStream_t stream, float* frame, float* bright, int W, int H, int k, float threshold, const BloomWeights& w)
{
    const size_t n = (size_t)W * H;
    const int blocks = (int)((n + 255) / 256);
    hipLaunchKernelGGL(bloomHighpassKernel, dim3(blocks), dim3(256), 0, stream, frame, bright, n, threshold);
    hipLaunchKernelGGL(bloomApplyKernel, dim3(blocks), dim3(256), 0, stream, frame, bright, W, H, k, w);
    return hipGetLastError();
}

hipError_t launchPackBmp(hipStream_t stream, const float* frame, uint8_t* out, int W, int H)
{
    const size_t n = (size_t)W * H;
    hipLaunchKernelGGL(packBmpKernel, dim3((int)((n + 255) / 256)), dim3(256), 0, stream, frame, out, W, H);
    return hipGetLastError();
}

} // namespace rt
