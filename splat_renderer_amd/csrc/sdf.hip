// sdf.hip — splat generation from a signed-distance scene: the producer of the positions and
// normals the tile-raster path consumes.
//
// Reference: /root/reference/src/sdf/CodeGenerator.ts:97-225 (primitive and operation library: sdgSphere, sdgBox,
// sdgTorus, sdgCapsule; opUnion, opIntersection, opSubtraction, opSmoothUnion), :276-353 (sceneSDF: post-order walk
// of the scene graph), src/GradientSampler.ts (K: gradients[i] = sceneSDF(positions[i])),
// src/shaders/update-positions.wgsl:22-50 (project a point onto the surface along its gradient),
// src/CurvatureSampler.ts:84-141 (six jittered normals -> scale factor).
//
// The reference GENERATES a WGSL function per scene graph and recompiles when the structure changes
// (CurvatureSampler.ts:169-191).  Here the scene graph travels as data: a postfix program of at most
// SPLAT_SDF_MAX_INSTR instructions in the kernel arguments (constant memory for every lane), evaluated by a small
// stack machine — no runtime compiler in the frame loop, and an animated parameter (src/main.ts:114-116) is just
// new kernel arguments.  A value is vec4(distance, gradient) as in the reference.
//
// Compiled with -ffp-contract=off: one IEEE binary32 operation per operator in the order written, identical to
// oracle/oracle.c (orc_sdf_*), so gradients, positions and scale factors are bit-exact against it.  WGSL built-ins are
// spelled out the same way in both: length(v) = sqrt((x*x + y*y) + z*z), normalize(v) = v / length(v),
// mix(a, b, t) = a * (1 - t) + b * t, smoothstep(lo, hi, x) = t*t*(3 - 2t) with t = clamp((x - lo) / (hi - lo), 0, 1).
#include "common.h"

struct SdfProgram {
    splat_sdf_instr instr[SPLAT_SDF_MAX_INSTR];
    uint32_t count;
};

constexpr int SDF_STACK = 8;

__device__ __forceinline__ float sdf_len3(float x, float y, float z) { return sqrtf((x * x + y * y) + z * z); }
__device__ __forceinline__ float sdf_len2(float x, float y) { return sqrtf(x * x + y * y); }
__device__ __forceinline__ float sdf_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

// CodeGenerator.ts:100-106
__device__ __forceinline__ float4 sdg_sphere(float px, float py, float pz, float r) {
    const float d = sdf_len3(px, py, pz);
    const float m = fmaxf(d, 0.0001f);
    return make_float4(d - r, px / m, py / m, pz / m);
}

// CodeGenerator.ts:109-133
__device__ __forceinline__ float4 sdg_box(float px, float py, float pz, float bx, float by, float bz) {
    const float qx = fabsf(px) - bx, qy = fabsf(py) - by, qz = fabsf(pz) - bz;
    const float wx = fmaxf(qx, 0.0f), wy = fmaxf(qy, 0.0f), wz = fmaxf(qz, 0.0f);
    const float g = fmaxf(qx, fmaxf(qy, qz));
    const float dist = sdf_len3(wx, wy, wz) + fminf(g, 0.0f);
    const float sx = sdf_sign(px), sy = sdf_sign(py), sz = sdf_sign(pz);
    float gx, gy, gz;
    if (g > 0.0f) {
        const float l = sdf_len3(wx, wy, wz); // normalize(w)
        gx = sx * (wx / l);
        gy = sy * (wy / l);
        gz = sz * (wz / l);
    } else if (qx > qy && qx > qz) { // inside: towards the nearest face
        gx = sx; gy = 0.0f; gz = 0.0f;
    } else if (qy > qz) {
        gx = 0.0f; gy = sy; gz = 0.0f;
    } else {
        gx = 0.0f; gy = 0.0f; gz = sz;
    }
    return make_float4(dist, gx, gy, gz);
}

// CodeGenerator.ts:136-157 (t = (major, minor))
__device__ __forceinline__ float4 sdg_torus(float px, float py, float pz, float major, float minor) {
    const float lxz = sdf_len2(px, pz);
    const float dx = lxz - major, dy = py;
    const float ldir = sdf_len2(dx, dy);
    const float dist = ldir - minor;
    float gx = 0.0f, gy = 1.0f, gz = 0.0f;
    if (lxz > 0.0001f && ldir > 0.0001f) {
        const float ux = px / lxz, uz = pz / lxz;
        const float ddx = dx / ldir, ddy = dy / ldir;
        gx = ux * ddx;
        gy = ddy;
        gz = uz * ddx;
    }
    return make_float4(dist, gx, gy, gz);
}

// CodeGenerator.ts:160-176
__device__ __forceinline__ float4 sdg_capsule(float px, float py, float pz, float h, float r) {
    const float half = h * 0.5f;
    const float cy = fminf(fmaxf(py, -half), half);
    const float qx = px, qy = py - cy, qz = pz;
    const float d = sdf_len3(qx, qy, qz);
    float gx = 0.0f, gy = sdf_sign(py), gz = 0.0f;
    if (d > 0.0001f) {
        gx = qx / d;
        gy = qy / d;
        gz = qz / d;
    }
    return make_float4(d - r, gx, gy, gz);
}

// CodeGenerator.ts:206-224
__device__ __forceinline__ float4 op_smooth_union(float4 a, float4 b, float k) {
    const float k4 = k * 4.0f;
    const float diff = fabsf(a.x - b.x);
    const float h = fmaxf(k4 - diff, 0.0f) / k4;
    const float dist = fminf(a.x, b.x) - ((h * h) * k4) * 0.25f;
    const float hg = fmaxf(k4 - diff, 0.0f) / (2.0f * k4);
    const float t = (a.x < b.x) ? hg : (1.0f - hg); // select(1 - hGrad, hGrad, a.x < b.x)
    const float u = 1.0f - t;                        // mix(a, b, t) = a * (1 - t) + b * t
    return make_float4(dist, a.y * u + b.y * t, a.z * u + b.z * t, a.w * u + b.w * t);
}

// sceneSDF (CodeGenerator.ts:276-353): the post-order walk of the scene graph, as a stack machine.
// The operand stack lives in LDS, one column per lane (`stack` = the lane's slot of level 0, levels SDF_LANES apart): as a
// local array it is indexed by a run-time stack pointer and the compiler spills it to scratch — 144 bytes per lane of
// global memory behind every push and pop, which made the producer kernels several times slower than their arithmetic.
constexpr int SDF_LANES = 64; // threads per workgroup of every kernel that evaluates the scene
#define SDF_STACK_DECL __shared__ float4 s_sdf_stack[SDF_STACK * SDF_LANES]; float4 *sdf_stack = s_sdf_stack + threadIdx.x
__device__ __forceinline__ float4 scene_sdf(const SdfProgram &prog, float px, float py, float pz, float4 *stack) {
    int sp = 0;
    for (uint32_t k = 0; k < prog.count; ++k) {
        const splat_sdf_instr &in = prog.instr[k];
        if (in.op < SPLAT_SDF_UNION) {
            const float x = px - in.a[0], y = py - in.a[1], z = pz - in.a[2]; // p - center
            float4 v;
            if (in.op == SPLAT_SDF_SPHERE) v = sdg_sphere(x, y, z, in.a[3]);
            else if (in.op == SPLAT_SDF_BOX) v = sdg_box(x, y, z, in.a[3], in.a[4], in.a[5]);
            else if (in.op == SPLAT_SDF_TORUS) v = sdg_torus(x, y, z, in.a[3], in.a[4]);
            else v = sdg_capsule(x, y, z, in.a[3], in.a[4]);
            stack[(sp++) * SDF_LANES] = v;
        } else {
            const float4 b = stack[(--sp) * SDF_LANES], a = stack[(--sp) * SDF_LANES];
            float4 v;
            if (in.op == SPLAT_SDF_UNION) v = (a.x < b.x) ? a : b;             // :181-187
            else if (in.op == SPLAT_SDF_INTERSECTION) v = (a.x > b.x) ? a : b;  // :190-196
            else if (in.op == SPLAT_SDF_SUBTRACTION) {                          // :199-202
                const float4 nb = make_float4(-b.x, -b.y, -b.z, -b.w);
                v = (a.x > nb.x) ? a : nb;
            } else v = op_smooth_union(a, b, in.a[0]);
            stack[(sp++) * SDF_LANES] = v;
        }
    }
    if (sp == 0) return make_float4(1000.0f, 0.0f, 1.0f, 0.0f); // empty scene (:282-286)
    return stack[(sp - 1) * SDF_LANES];
}

// GradientSampler (CodeGenerator.ts:72-90)
__global__ __launch_bounds__(64) void k_sdf_gradients(SdfProgram prog, const float4 *__restrict__ positions, uint32_t n,
                                                      float4 *__restrict__ gradients) {
    SDF_STACK_DECL;
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const float4 p = positions[i];
    gradients[i] = scene_sdf(prog, p.x, p.y, p.z, sdf_stack);
}

// PositionUpdater (update-positions.wgsl:22-50)
__device__ __forceinline__ float4 sdf_step_position(float4 p, float4 g) {
    const float len = sdf_len3(g.y, g.z, g.w);
    float x = p.x, y = p.y, z = p.z;
    if (len > 0.0001f) { // :42-45: newPos = pos - normalize(grad) * distance
        x = p.x - (g.y / len) * g.x;
        y = p.y - (g.z / len) * g.x;
        z = p.z - (g.w / len) * g.x;
    }
    return make_float4(x, y, z, 0.0f);
}

__global__ __launch_bounds__(64) void k_sdf_update_positions(const float4 *__restrict__ positions, const float4 *__restrict__ gradients,
                                                             uint32_t n, float4 *__restrict__ next_positions) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    next_positions[i] = sdf_step_position(positions[i], gradients[i]);
}

// CurvatureSampler (CurvatureSampler.ts:84-141)
__device__ __forceinline__ float sdf_scale_factor(const SdfProgram &prog, float4 c, float4 *sdf_stack) {
    const float r = 0.02f; // sampleRadius
    const float4 cr = scene_sdf(prog, c.x, c.y, c.z, sdf_stack);
    const float cl = sdf_len3(cr.y, cr.z, cr.w);
    const float nx = cr.y / cl, ny = cr.z / cl, nz = cr.w / cl;
    float total = 0.0f;
    for (int k = 0; k < 6; ++k) { // the six axis offsets, in the reference's order (:100-107)
        const float ox = (k == 0) ? r : (k == 1) ? -r : 0.0f;
        const float oy = (k == 2) ? r : (k == 3) ? -r : 0.0f;
        const float oz = (k == 4) ? r : (k == 5) ? -r : 0.0f;
        const float4 s = scene_sdf(prog, c.x + ox, c.y + oy, c.z + oz, sdf_stack);
        const float sl = sdf_len3(s.y, s.z, s.w);
        const float d = (nx * (s.y / sl) + ny * (s.z / sl)) + nz * (s.w / sl);
        total = total + (1.0f - d); // :121-123
    }
    const float avg = total / 6.0f;
    const float t = fminf(fmaxf((avg - 0.0f) / (0.5f - 0.0f), 0.0f), 1.0f); // smoothstep(0, 0.5, avg) :131
    const float sm = (t * t) * (3.0f - 2.0f * t);
    const float sf = 1.0f - sm;
    return 0.01f * (1.0f - sf) + 1.0f * sf; // mix(0.01, 1.0, scaleFactor) :132
}

__global__ __launch_bounds__(64) void k_sdf_scale_factors(SdfProgram prog, const float4 *__restrict__ positions, uint32_t n,
                                                          float *__restrict__ scale_factors) {
    SDF_STACK_DECL;
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    scale_factors[i] = sdf_scale_factor(prog, positions[i], sdf_stack);
}

// The buffer SplatPropertyManager.updateFromCurvature binds as "curvatureData" is vec4(normal.xyz, scaleFactor)
// (src/SplatPropertyManager.ts:70-72) while the samplers above produce vec4(distance, gradient) and one f32 per point
// (SURVEY I4): this joins the two — normal = normalize(gradient), or (0, 1, 0) where the gradient vanishes.
__device__ __forceinline__ float4 sdf_curvature_of(float4 g, float scale_factor) {
    const float len = sdf_len3(g.y, g.z, g.w);
    float x = 0.0f, y = 1.0f, z = 0.0f;
    if (len > 0.0001f) {
        x = g.y / len;
        y = g.z / len;
        z = g.w / len;
    }
    return make_float4(x, y, z, scale_factor);
}

__global__ __launch_bounds__(64) void k_sdf_curvature(const float4 *__restrict__ gradients, const float *__restrict__ scale_factors,
                                                      uint32_t n, float4 *__restrict__ curvature) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    curvature[i] = sdf_curvature_of(gradients[i], scale_factors[i]);
}

// stack discipline of a postfix program, checked on the host before anything is launched
static int sdf_load(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, SdfProgram &out) {
    if (n_instr > SPLAT_SDF_MAX_INSTR) return ctx_fail(ctx, SPLAT_ERR_INVALID, "SDF program: more than SPLAT_SDF_MAX_INSTR instructions");
    if (n_instr && !program) return ctx_fail(ctx, SPLAT_ERR_INVALID, "SDF program is NULL");
    int depth = 0;
    for (uint32_t k = 0; k < n_instr; ++k) {
        const uint32_t op = program[k].op;
        if (op <= SPLAT_SDF_CAPSULE) {
            if (++depth > SDF_STACK) return ctx_fail(ctx, SPLAT_ERR_INVALID, "SDF program: scene graph deeper than the evaluator's stack (8)");
        } else if (op >= SPLAT_SDF_UNION && op <= SPLAT_SDF_SMOOTH_UNION) {
            if (depth < 2) return ctx_fail(ctx, SPLAT_ERR_INVALID, "SDF program: an operation without two operands");
            --depth;
        } else {
            return ctx_fail(ctx, SPLAT_ERR_INVALID, "SDF program: unknown opcode");
        }
        out.instr[k] = program[k];
    }
    if (n_instr && depth != 1) return ctx_fail(ctx, SPLAT_ERR_INVALID, "SDF program: does not reduce to one value");
    out.count = n_instr;
    return SPLAT_OK;
}

// ---- PointManager.generateRandomPositions on the device (splat.h: splat_sdf_seed_positions) ------------------------
struct SeedBox {
    float mn[3], mx[3];
};

__device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

__device__ __forceinline__ float4 sdf_seed_point(const SeedBox &box, uint64_t seed, uint32_t i) {
    const uint64_t base = seed * 0x9E3779B97F4A7C15ull + 2ull * i;
    const uint64_t a = splitmix64(base), b = splitmix64(base + 1ull);
    const float k24 = 1.0f / 16777216.0f; // 24 bits per uniform: exact in binary32, [0, 1)
    const float u0 = (float)(uint32_t)(a >> 40) * k24, u1 = (float)(uint32_t)((a >> 16) & 0xffffffu) * k24;
    const float u2 = (float)(uint32_t)(b >> 40) * k24, u3 = (float)(uint32_t)((b >> 16) & 0xffffffu) * k24;
    const float d[3] = {box.mx[0] - box.mn[0], box.mx[1] - box.mn[1], box.mx[2] - box.mn[2]};
    // faces -x, +x, -y, +y, -z, +z, each with its area (PointManager.ts:108-131); one rounding per operation
    const float ax = d[1] * d[2], ay = d[0] * d[2], az = d[0] * d[1];
    const float c0 = ax, c1 = c0 + ax, c2 = c1 + ay, c3 = c2 + ay, c4 = c3 + az, c5 = c4 + az;
    const float t = u0 * c5;
    const uint32_t face = t < c0 ? 0u : t < c1 ? 1u : t < c2 ? 2u : t < c3 ? 3u : t < c4 ? 4u : 5u;
    float x = box.mn[0] + u1 * d[0], y = box.mn[1] + u2 * d[1], z = box.mn[2] + u3 * d[2];
    // (selects, not arrays indexed by the face: those end up in scratch)
    if (face == 0) x = box.mn[0];
    else if (face == 1) x = box.mx[0];
    else if (face == 2) y = box.mn[1];
    else if (face == 3) y = box.mx[1];
    else if (face == 4) z = box.mn[2];
    else z = box.mx[2];
    return make_float4(x, y, z, 0.0f);
}

__global__ __launch_bounds__(256) void k_sdf_seed_positions(SeedBox box, uint32_t n, uint64_t seed, float4 *__restrict__ positions) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n) positions[i] = sdf_seed_point(box, seed, i);
}

// The producer half of the reference's frame in ONE launch (src/main.ts:146-180): a point's fresh position (seeded here, or
// read), `steps` rounds of {evaluate the scene at it, step onto the surface}, the curvature scale factor at where it ends
// up, vec4(normal of the LAST evaluation, scale) and the splat's property record — the same device functions, in the same
// order, as the stage kernels above (and k_update_props), so every output has the bits theirs have.  A point never looks
// at another, so nothing is lost by not materialising the rounds; what is gained is thirteen launches of ~6 us each on
// 10^5 points, which is what the stage-by-stage form of this producer costs (0.086 ms of a 0.30 ms frame at the
// reference's working point).
__global__ __launch_bounds__(64) void k_sdf_generate(SdfProgram prog, SeedBox box, uint32_t seeded, uint64_t seed,
                                                     const float4 *__restrict__ positions_in, uint32_t n, uint32_t steps,
                                                     float4 *__restrict__ positions_out, float4 *__restrict__ gradients_out,
                                                     float4 *__restrict__ curvature_out, float4 *__restrict__ props_out) {
    SDF_STACK_DECL;
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    float4 p = seeded ? sdf_seed_point(box, seed, i) : positions_in[i];
    float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (uint32_t k = 0; k < steps; ++k) {
        g = scene_sdf(prog, p.x, p.y, p.z, sdf_stack);
        p = sdf_step_position(p, g);
    }
    const float4 c = sdf_curvature_of(g, sdf_scale_factor(prog, p, sdf_stack));
    positions_out[i] = p;
    if (gradients_out) gradients_out[i] = g;
    curvature_out[i] = c;
    if (props_out) { // SplatPropertyManager.updateFromCurvature (src/SplatPropertyManager.ts:82-107), as k_update_props
        props_out[(size_t)i * 2] = make_float4(p.x, p.y, p.z, 0.04f);                                                   // :94
        props_out[(size_t)i * 2 + 1] = make_float4(fabsf(c.x) * 0.8f + 0.2f, fabsf(c.y) * 0.8f + 0.2f, fabsf(c.z) * 0.8f + 0.2f, 1.0f); // :97-101
    }
}

extern "C" {

int splat_sdf_gradients(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, const void *positions, uint32_t n,
                        void *gradients) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (positions && gradients));
    ARG_CHECK(ctx, (((uintptr_t)positions | (uintptr_t)gradients) & 15) == 0);
    SdfProgram prog;
    int rc = sdf_load(ctx, program, n_instr, prog);
    if (rc != SPLAT_OK || n == 0) return rc;
    hipLaunchKernelGGL(k_sdf_gradients, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, prog, (const float4 *)positions, n, (float4 *)gradients);
    LAUNCH_CHECK(ctx, "k_sdf_gradients");
    return SPLAT_OK;
}

int splat_sdf_update_positions(splat_ctx *ctx, const void *positions, const void *gradients, uint32_t n, void *next_positions) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (positions && gradients && next_positions));
    ARG_CHECK(ctx, (((uintptr_t)positions | (uintptr_t)gradients | (uintptr_t)next_positions) & 15) == 0);
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_sdf_update_positions, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, (const float4 *)positions,
                       (const float4 *)gradients, n, (float4 *)next_positions);
    LAUNCH_CHECK(ctx, "k_sdf_update_positions");
    return SPLAT_OK;
}

int splat_sdf_scale_factors(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, const void *positions, uint32_t n,
                            void *scale_factors) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (positions && scale_factors));
    ARG_CHECK(ctx, ((uintptr_t)positions & 15) == 0 && ((uintptr_t)scale_factors & 3) == 0);
    SdfProgram prog;
    int rc = sdf_load(ctx, program, n_instr, prog);
    if (rc != SPLAT_OK || n == 0) return rc;
    hipLaunchKernelGGL(k_sdf_scale_factors, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, prog, (const float4 *)positions, n,
                       (float *)scale_factors);
    LAUNCH_CHECK(ctx, "k_sdf_scale_factors");
    return SPLAT_OK;
}

int splat_sdf_curvature(splat_ctx *ctx, const void *gradients, const void *scale_factors, uint32_t n, void *curvature) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, n == 0 || (gradients && scale_factors && curvature));
    ARG_CHECK(ctx, (((uintptr_t)gradients | (uintptr_t)curvature) & 15) == 0);
    if (n == 0) return SPLAT_OK;
    hipLaunchKernelGGL(k_sdf_curvature, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, (const float4 *)gradients,
                       (const float *)scale_factors, n, (float4 *)curvature);
    LAUNCH_CHECK(ctx, "k_sdf_curvature");
    return SPLAT_OK;
}

int splat_sdf_seed_positions(splat_ctx *ctx, const float *aabb_min3, const float *aabb_max3, uint32_t n, uint64_t seed,
                             void *positions) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    ARG_CHECK(ctx, aabb_min3 && aabb_max3 && (n == 0 || positions));
    ARG_CHECK(ctx, (((uintptr_t)positions) & 15) == 0);
    if (n == 0) return SPLAT_OK;
    SeedBox box;
    for (int a = 0; a < 3; ++a) {
        box.mn[a] = aabb_min3[a];
        box.mx[a] = aabb_max3[a];
    }
    hipLaunchKernelGGL(k_sdf_seed_positions, dim3(div_up(n, 256)), dim3(256), 0, ctx->stream, box, n, seed, (float4 *)positions);
    LAUNCH_CHECK(ctx, "k_sdf_seed_positions");
    return SPLAT_OK;
}

int splat_sdf_generate(splat_ctx *ctx, const splat_sdf_instr *program, uint32_t n_instr, const float *aabb_min3, const float *aabb_max3,
                       uint64_t seed, const void *positions_in, uint32_t n, uint32_t steps, void *positions_out, void *gradients_out,
                       void *curvature_out, void *props_out) {
    if (!ctx) return ctx_fail(nullptr, SPLAT_ERR_INVALID, "ctx is NULL");
    const bool seeded = aabb_min3 != nullptr;
    ARG_CHECK(ctx, seeded == (aabb_max3 != nullptr) && (n == 0 || seeded || positions_in));
    ARG_CHECK(ctx, n == 0 || (positions_out && curvature_out));
    ARG_CHECK(ctx, steps >= 1); // (the normal is that of the last evaluation: there has to be one)
    ARG_CHECK(ctx, (((uintptr_t)positions_in | (uintptr_t)positions_out | (uintptr_t)gradients_out | (uintptr_t)curvature_out |
                     (uintptr_t)props_out) & 15) == 0);
    SdfProgram prog;
    int rc = sdf_load(ctx, program, n_instr, prog);
    if (rc != SPLAT_OK) return rc;
    if (n == 0) return SPLAT_OK;
    SeedBox box = {};
    if (seeded)
        for (int a = 0; a < 3; ++a) {
            box.mn[a] = aabb_min3[a];
            box.mx[a] = aabb_max3[a];
        }
    hipLaunchKernelGGL(k_sdf_generate, dim3(div_up(n, 64)), dim3(64), 0, ctx->stream, prog, box, seeded ? 1u : 0u, seed,
                       (const float4 *)positions_in, n, steps, (float4 *)positions_out, (float4 *)gradients_out, (float4 *)curvature_out,
                       (float4 *)props_out);
    LAUNCH_CHECK(ctx, "k_sdf_generate");
    return SPLAT_OK;
}

} // extern "C"
