// On-device input pipeline: the image half of load_image_labels (reference
// utils/data/__init__.py:162-175) after JPEG decode -- crop_to_bounding_box -> resize_images (bilinear) -> flip ->
// brightness -> saturation -> hue -> contrast -> noise -> grayscale -> clip -- and transform_labels
// (utils/data/__init__.py:112-145), so a training step consumes decoded uint8 images and raw boxes resident in HBM
// instead of a host-side TF queue.
//
// Every random decision of the reference (tf.random_uniform / tf.cond / truncated_normal's scale) is drawn on the
// host and arrives in yolo2_augment_params; only the per-pixel noise is generated here (counter-based hash ->
// Box-Muller, resampled to |z| <= 2 like tf.truncated_normal).  The image arithmetic restates the TF-1.0 kernels in
// f32 in their operation order (ResizeBilinear align_corners=False, RGBToHSV / HSVToRGB, AdjustContrastv2,
// rgb_to_grayscale weights); oracle/yolo2_ref.py holds the same restatement (parity unpinned by the reference: the
// arithmetic lives in TensorFlow).  HBM-bound: one uint8 source read (4 taps, L2-resident) + one f32 write per pixel;
// contrast needs the per-channel image mean of the value BEFORE contrast, so images that take that branch are
// evaluated twice (pass A: sums, pass B: output) rather than staged.
#include "common.h"
#pragma clang fp contract(off)

struct Px { float r, g, b; };

__device__ __forceinline__ Px rgb_to_hsv_tf(Px p) {
    const float v = fmaxf(fmaxf(p.r, p.g), p.b);
    const float range = v - fminf(fminf(p.r, p.g), p.b);
    const float s = v > 0.f ? range / v : 0.f;
    const float norm = 1.0f / (6.0f * range);
    float h;
    if (p.r == v) h = norm * (p.g - p.b);
    else if (p.g == v) h = norm * (p.b - p.r) + (float)(2.0 / 6.0);
    else h = norm * (p.r - p.g) + (float)(4.0 / 6.0);
    if (range <= 0.f) h = 0.f;
    if (h < 0.f) h = h + 1.0f;
    return Px{h, s, v};
}

__device__ __forceinline__ Px hsv_to_rgb_tf(Px q) {
    const float c = q.g * q.b;      // s * v
    const float m = q.b - c;
    const float dh = q.r * 6.0f;
    float fmodu = dh;
    while (fmodu <= 0.f) fmodu += 2.0f;
    while (fmodu >= 2.0f) fmodu -= 2.0f;
    const float x = c * (1.0f - fabsf(fmodu - 1.0f));
    const int cat = (int)dh;
    float rr = 0.f, gg = 0.f, bb = 0.f;
    switch (cat) {
        case 0: rr = c; gg = x; break;
        case 1: rr = x; gg = c; break;
        case 2: gg = c; bb = x; break;
        case 3: gg = x; bb = c; break;
        case 4: rr = x; bb = c; break;
        case 5: rr = c; bb = x; break;
        default: break;
    }
    return Px{rr + m, gg + m, bb + m};
}

// value of output pixel (y, x) after crop, resize, flip, brightness, saturation, hue (everything before contrast)
__device__ __forceinline__ Px pixel_before_contrast(const unsigned char *__restrict__ src, const yolo2_augment_params &p, int y, int x, int H, int W) {
    const unsigned char *img = src + p.src_offset;
    if (p.flags & YOLO2_AUG_FLIP) x = W - 1 - x;
    Px v;
    const long row = (long)p.src_w * 3;
    if (p.crop_h == H && p.crop_w == W) {          // resize_images returns the input unchanged when the size matches
        const unsigned char *t = img + (long)(p.crop_y + y) * row + (long)(p.crop_x + x) * 3;
        v = Px{(float)t[0], (float)t[1], (float)t[2]};
    } else {
        const float hs = (float)p.crop_h / (float)H, ws = (float)p.crop_w / (float)W;
        const float fy = (float)y * hs, fx = (float)x * ws;
        const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const int y1 = min((int)ceilf(fy), p.crop_h - 1), x1 = min((int)ceilf(fx), p.crop_w - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0;
        const unsigned char *r0 = img + (long)(p.crop_y + y0) * row + (long)p.crop_x * 3;
        const unsigned char *r1 = img + (long)(p.crop_y + y1) * row + (long)p.crop_x * 3;
        float o[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float tl = (float)r0[x0 * 3 + c], tr = (float)r0[x1 * 3 + c];
            const float bl = (float)r1[x0 * 3 + c], br = (float)r1[x1 * 3 + c];
            const float top = tl + (tr - tl) * lx;
            const float bottom = bl + (br - bl) * lx;
            o[c] = top + (bottom - top) * ly;
        }
        v = Px{o[0], o[1], o[2]};
    }
    if (p.flags & YOLO2_AUG_BRIGHTNESS) { v.r += p.brightness; v.g += p.brightness; v.b += p.brightness; }
    if (p.flags & YOLO2_AUG_SATURATION) {
        Px q = rgb_to_hsv_tf(v);
        q.g = fminf(fmaxf(q.g * p.saturation, 0.f), 1.f);
        v = hsv_to_rgb_tf(q);
    }
    if (p.flags & YOLO2_AUG_HUE) {
        Px q = rgb_to_hsv_tf(v);
        float h = q.r + (1.0f + p.hue);
        h = h - floorf(h);                           // floormod(h, 1)
        q.r = h;
        v = hsv_to_rgb_tf(q);
    }
    return v;
}

// pass A: per image, per channel sums of the pre-contrast value (only images with the contrast flag do work)
__global__ __launch_bounds__(256) void augment_sums_kernel(const unsigned char *__restrict__ src, const yolo2_augment_params *__restrict__ params,
                                                           double *__restrict__ sums, int H, int W) {
    const int b = blockIdx.y;
    const yolo2_augment_params p = params[b];
    if (!(p.flags & YOLO2_AUG_CONTRAST)) return;
    double s[3] = {0.0, 0.0, 0.0};
    const int total = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const Px v = pixel_before_contrast(src, p, i / W, i % W, H, W);
        s[0] += (double)v.r;
        s[1] += (double)v.g;
        s[2] += (double)v.b;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double t = wave_sum_d(s[c]);
        if ((threadIdx.x & 63) == 0) atomicAdd(sums + 3 * b + c, t);
    }
}

__device__ __forceinline__ unsigned hash32(unsigned x) {       // "lowbias32" integer finaliser
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// Three standard normals truncated to |z| <= 2 by resampling (tf.truncated_normal), counter-based on (seed, pixel):
// two Box-Muller pairs per pixel from four 24-bit uniforms; a component beyond 2 sigma (4.6 %) is redrawn on its own
// counter stream.  Fast-math log / sincos: this is noise, not parity arithmetic.
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float &z0, float &z1) {
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);      // (0, 1]
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);               // [0, 1)
    const float r = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.28318530717958647692f * u2, &sn, &cs);
    z0 = r * cs;
    z1 = r * sn;
}
__device__ __forceinline__ void truncated_normal3(unsigned long long seed, unsigned pixel, float (&z)[3]) {
    const unsigned s0 = (unsigned)seed, s1 = (unsigned)(seed >> 32);
    const unsigned base = hash32(pixel ^ s0) + s1;
    float t[4];
    box_muller(hash32(base), hash32(base + 0x9e3779b9u), t[0], t[1]);
    box_muller(hash32(base + 0x3c6ef372u), hash32(base + 0xdaa66d2bu), t[2], t[3]);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v = t[c];
        if (fabsf(v) > 2.0f) {
            v = fabsf(t[3]) <= 2.0f && c == 0 ? t[3] : 3.0f;           // the spare fourth sample serves the first reject
            unsigned ctr = hash32(base ^ (0x85ebca6bu * (unsigned)(c + 1)));
            for (int attempt = 0; attempt < 32 && fabsf(v) > 2.0f; ++attempt) {
                float w;
                box_muller(hash32(ctr), hash32(ctr + 0x9e3779b9u), v, w);
                if (fabsf(v) > 2.0f) v = w;
                ctr += 0x632be5abu;
            }
            if (fabsf(v) > 2.0f) v = 0.f;
        }
        z[c] = v;
    }
}

// pass B: the whole chain, f32 [B, H, W, 3] out
__global__ __launch_bounds__(256) void augment_apply_kernel(const unsigned char *__restrict__ src, const yolo2_augment_params *__restrict__ params,
                                                            const double *__restrict__ sums, float *__restrict__ out, int H, int W) {
    const int b = blockIdx.y;
    const yolo2_augment_params p = params[b];
    const int total = H * W;
    float mean[3] = {0.f, 0.f, 0.f};
    if (p.flags & YOLO2_AUG_CONTRAST) {
#pragma unroll
        for (int c = 0; c < 3; ++c) mean[c] = (float)(sums[3 * b + c] / (double)total);
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        Px v = pixel_before_contrast(src, p, i / W, i % W, H, W);
        if (p.flags & YOLO2_AUG_CONTRAST) {
            v.r = (v.r - mean[0]) * p.contrast + mean[0];
            v.g = (v.g - mean[1]) * p.contrast + mean[1];
            v.b = (v.b - mean[2]) * p.contrast + mean[2];
        }
        if (p.flags & YOLO2_AUG_NOISE) {
            float z[3];
            truncated_normal3(p.noise_seed, (unsigned)i, z);
            v.r = v.r + z[0] * p.noise_scale;
            v.g = v.g + z[1] * p.noise_scale;
            v.b = v.b + z[2] * p.noise_scale;
        }
        if (p.flags & YOLO2_AUG_GRAY) {
            const float g = (v.r * 0.2989f + v.g * 0.5870f) + v.b * 0.1140f;
            v = Px{g, g, g};
        }
        float *o = out + ((long)b * total + i) * 3;
        o[0] = fminf(fmaxf(v.r, 0.f), 255.f);
        o[1] = fminf(fmaxf(v.g, 0.f), 255.f);
        o[2] = fminf(fmaxf(v.b, 0.f), 255.f);
    }
}

extern "C" int yolo2_augment_images(const unsigned char *src, const yolo2_augment_params *params_device, double *ws, float *out, int B, int H,
                                    int W, int any_contrast, void *stream) {
    Y2_CHECK_ARG(src && params_device && out && B > 0 && H > 0 && W > 0);
    Y2_CHECK_ARG(!any_contrast || ws);
    hipStream_t st = (hipStream_t)stream;
    const int bx = B >= 64 ? 8 : B >= 8 ? 32 : 128;
    dim3 grid(bx, B);
    if (any_contrast) {
        if (hipMemsetAsync(ws, 0, sizeof(double) * 3 * B, st) != hipSuccess) { yolo2_set_error("yolo2_augment_images: memset failed"); return YOLO2_E_LAUNCH; }
        augment_sums_kernel<<<grid, 256, 0, st>>>(src, params_device, ws, H, W);
    }
    augment_apply_kernel<<<grid, 256, 0, st>>>(src, params_device, ws, out, H, W);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}

// ---------------------------------------------------------------------------------------------------------
// transform_labels (utils/data/__init__.py:112-145): boxes normalised to [0,1] -> the six label tensors of the loss.
// One workgroup per image: all threads clear the image's slices, then one lane replays the objects IN ORDER (NumPy
// fancy assignment with repeated indices: the last object of a cell wins for coords / offsets, class bits accumulate),
// f32 arithmetic in NumPy's operation order with contraction off -> bit-exact against the reference's goldens.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transform_labels_kernel(const int *__restrict__ cls, const float *__restrict__ box, const int *__restrict__ first,
                                                               float *__restrict__ mask, float *__restrict__ prob, float *__restrict__ coords,
                                                               float *__restrict__ off_min, float *__restrict__ off_max, float *__restrict__ areas,
                                                               int classes, int cell_w, int cell_h, int *__restrict__ err) {
    const int b = blockIdx.x, cells = cell_w * cell_h;
    float *m = mask + (long)b * cells, *pr = prob + (long)b * cells * classes, *co = coords + (long)b * cells * 4;
    float *mn = off_min + (long)b * cells * 2, *mx = off_max + (long)b * cells * 2, *ar = areas + (long)b * cells;
    for (int i = threadIdx.x; i < cells; i += blockDim.x) m[i] = 0.f;
    for (int i = threadIdx.x; i < cells * classes; i += blockDim.x) pr[i] = 0.f;
    for (int i = threadIdx.x; i < cells * 4; i += blockDim.x) co[i] = 0.f;
    for (int i = threadIdx.x; i < cells * 2; i += blockDim.x) { mn[i] = 0.f; mx[i] = 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float cw = (float)cell_w, ch = (float)cell_h;
        for (int k = first[b]; k < first[b + 1]; ++k) {
            const float xmin = box[4 * k], ymin = box[4 * k + 1], xmax = box[4 * k + 2], ymax = box[4 * k + 3];
            const float x = cw * (xmin + xmax) / 2.0f, y = ch * (ymin + ymax) / 2.0f;
            const float ix = floorf(x), iy = floorf(y);
            const float ox = x - ix, oy = y - iy;
            const float w = xmax - xmin, h = ymax - ymin;
            const int index = (int)(iy * cw + ix);
            const int c = cls[k];
            if (index < 0 || index >= cells || c < 0 || c >= classes) { atomicOr(err, 1); continue; }   // the reference raises IndexError
            m[index] = 1.f;
            pr[(long)index * classes + c] = 1.f;
            co[index * 4 + 0] = ox;
            co[index * 4 + 1] = oy;
            co[index * 4 + 2] = (float)sqrt((double)w);      // correctly rounded f32 square root (np.sqrt): via f64, 53 >= 2*24+2
            co[index * 4 + 3] = (float)sqrt((double)h);
            const float hw = w / 2.0f * cw, hh = h / 2.0f * ch;
            mn[index * 2 + 0] = ox - hw;
            mn[index * 2 + 1] = oy - hh;
            mx[index * 2 + 0] = ox + hw;
            mx[index * 2 + 1] = oy + hh;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < cells; i += blockDim.x) {
        const float ew = mx[i * 2] - mn[i * 2], eh = mx[i * 2 + 1] - mn[i * 2 + 1];
        if (ew < 0.f || eh < 0.f) atomicOr(err, 2);          // the reference asserts wh >= 0
        ar[i] = ew * eh;
    }
}

extern "C" int yolo2_transform_labels(const int *objects_class, const float *objects_coord, const int *first_object, float *mask, float *prob,
                                      float *coords, float *offset_xy_min, float *offset_xy_max, float *areas, int B, int classes, int cell_width,
                                      int cell_height, int *error_flag, void *stream) {
    Y2_CHECK_ARG(objects_class && objects_coord && first_object && mask && prob && coords && offset_xy_min && offset_xy_max && areas && error_flag);
    Y2_CHECK_ARG(B > 0 && classes > 0 && cell_width > 0 && cell_height > 0);
    transform_labels_kernel<<<B, 256, 0, (hipStream_t)stream>>>(objects_class, objects_coord, first_object, mask, prob, coords, offset_xy_min,
                                                                offset_xy_max, areas, classes, cell_width, cell_height, error_flag);
    Y2_CHECK_LAUNCH();
    return YOLO2_OK;
}
