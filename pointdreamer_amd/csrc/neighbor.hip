// SURVEY 8(f)-2, complete_unseen_by='neighbor' (pointdreamer/unproject.py:93-196, demo.py:180-200): colour the texels no view
// painted from their mesh neighbours.  The mesh work that the reference also does on the host (two rounds of midpoint
// subdivision of the unpainted faces, utils/mesh_utils.py:7-114) stays on the host (pointdreamer_amd/mesh_utils.py); the
// per-vertex and per-texel work runs here:
//   k_mark_unpainted_faces   faces that own an unpainted chart texel               (demo.py:180-181)
//   k_vertex_texel_fetch     vertex -> texel, colour, has-colour                    (unproject.py:130-139)
//   k_nb_gather / k_nb_apply one Jacobi round of the (L + I) neighbour average      (unproject.py:160-166)
//   k_vc_owner / k_vc_write  vertex colours back into the atlas                     (unproject.py:187-188)
// followed by the exact nearest fill of nearest.hip (unproject.py:191-193).
// Arithmetic: float32, one rounding per op, ascending neighbour order (this unit is compiled with -ffp-contract=off), so the
// numpy oracle (oracle/neighbor.py) is reproduced bit for bit.  Duplicate writes: largest vertex index wins.
#include "common.h"
using namespace pdhip;

__global__ void k_mark_unpainted_faces(const int64_t* __restrict__ face_id, const uint8_t* __restrict__ painted, long long n,
                                       int F, uint8_t* __restrict__ flags) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int64_t f = face_id[i];
        if (f >= 0 && f < F && !painted[i]) flags[f] = 1;
    }
}

extern "C" int pdhip_mark_unpainted_faces(const int64_t* face_id, const uint8_t* painted, int A, int F, uint8_t* flags, void* stream) {
    PD_REQUIRE(face_id && painted && flags && A > 0 && F > 0, "pdhip_mark_unpainted_faces: bad arguments");
    hipStream_t s = as_stream(stream);
    PD_HIP(hipMemsetAsync(flags, 0, (size_t)F, s));
    const long long n = (long long)A * A;
    k_mark_unpainted_faces<<<min(cdiv(n, 256), 4096), 256, 0, s>>>(face_id, painted, n, F, flags);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__global__ void k_vertex_texel_fetch(const float* __restrict__ vert_uvs, int V, const float* __restrict__ atlas,
                                     const uint8_t* __restrict__ mask, int A, int32_t* __restrict__ texel,
                                     float* __restrict__ colors, float* __restrict__ count) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
        const int col = clip_to_int(vert_uvs[2 * v] * (float)A, A - 1);
        const int row = clip_to_int(vert_uvs[2 * v + 1] * (float)A, A - 1);
        texel[2 * v] = row; texel[2 * v + 1] = col;
        const size_t p = (size_t)row * A + col;
        colors[3 * v] = atlas[3 * p]; colors[3 * v + 1] = atlas[3 * p + 1]; colors[3 * v + 2] = atlas[3 * p + 2];
        count[v] = mask[p] ? 1.0f : 0.0f;
    }
}

extern "C" int pdhip_vertex_texel_fetch(const float* vert_uvs, int V, const float* atlas, const uint8_t* mask, int A, int32_t* texel,
                                        float* colors, float* count, void* stream) {
    PD_REQUIRE(vert_uvs && atlas && mask && texel && colors && count && V > 0 && A > 0, "pdhip_vertex_texel_fetch: bad arguments");
    k_vertex_texel_fetch<<<min(cdiv(V, 256), 2048), 256, 0, as_stream(stream)>>>(vert_uvs, V, atlas, mask, A, texel, colors, count);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

// new colour / weight of every invalid vertex from the OLD colours (Jacobi): tmp[a] = (sum w c_j n_j, sum w n_j), w = 1/deg
__global__ void k_nb_gather(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ colidx,
                            const int32_t* __restrict__ invalid, int IV, const float* __restrict__ colors,
                            const float* __restrict__ count, float* __restrict__ tmp) {
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < IV; a += gridDim.x * blockDim.x) {
        const int i = invalid[a];
        const int b = rowptr[i], e = rowptr[i + 1];
        float c0 = 0.f, c1 = 0.f, c2 = 0.f, n = 0.f;
        if (e > b) {
            const float w = 1.0f / (float)(e - b);
            for (int k = b; k < e; ++k) {
                const int j = colidx[k];
                const float nj = count[j];
                c0 = c0 + w * (colors[3 * j] * nj);
                c1 = c1 + w * (colors[3 * j + 1] * nj);
                c2 = c2 + w * (colors[3 * j + 2] * nj);
                n = n + w * nj;
            }
        }
        tmp[4 * a] = c0; tmp[4 * a + 1] = c1; tmp[4 * a + 2] = c2; tmp[4 * a + 3] = n;
    }
}

__global__ void k_nb_apply(const int32_t* __restrict__ invalid, int IV, const float* __restrict__ tmp, float* __restrict__ colors,
                           float* __restrict__ count, int* __restrict__ colored) {
    int mine = 0;
    for (int a = blockIdx.x * blockDim.x + threadIdx.x; a < IV; a += gridDim.x * blockDim.x) {
        const int i = invalid[a];
        const float n = tmp[4 * a + 3];
        if (n > 0.f) {
            colors[3 * i] = tmp[4 * a] / n; colors[3 * i + 1] = tmp[4 * a + 1] / n; colors[3 * i + 2] = tmp[4 * a + 2] / n;
            count[i] = 1.0f;
            ++mine;
        } else {
            count[i] = 0.0f;
        }
    }
    if (mine) atomicAdd(colored, mine);
}

extern "C" int pdhip_neighbor_diffuse_round(const int32_t* rowptr, const int32_t* colidx, int V, const int32_t* invalid, int IV,
                                            float* colors, float* count, float* tmp /*[IV*4]*/, int* colored /*device int*/,
                                            void* stream) {
    PD_REQUIRE(rowptr && colidx && invalid && colors && count && tmp && colored && V > 0 && IV >= 0,
               "pdhip_neighbor_diffuse_round: bad arguments");
    hipStream_t s = as_stream(stream);
    PD_HIP(hipMemsetAsync(colored, 0, sizeof(int), s));
    if (IV > 0) {
        const int grid = min(cdiv(IV, 256), 2048);
        k_nb_gather<<<grid, 256, 0, s>>>(rowptr, colidx, invalid, IV, colors, count, tmp);
        k_nb_apply<<<grid, 256, 0, s>>>(invalid, IV, tmp, colors, count, colored);
    }
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}

__global__ void k_vc_owner(const int32_t* __restrict__ texel, int V, int A, int32_t* __restrict__ owner) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x)
        atomicMax(&owner[(size_t)texel[2 * v] * A + texel[2 * v + 1]], v);
}
__global__ void k_vc_write(const int32_t* __restrict__ texel, const float* __restrict__ colors, int V, int A,
                           const int32_t* __restrict__ owner, float* __restrict__ atlas, uint8_t* __restrict__ mask) {
    for (int v = blockIdx.x * blockDim.x + threadIdx.x; v < V; v += gridDim.x * blockDim.x) {
        const size_t p = (size_t)texel[2 * v] * A + texel[2 * v + 1];
        if (owner[p] != v) continue;
        atlas[3 * p] = colors[3 * v]; atlas[3 * p + 1] = colors[3 * v + 1]; atlas[3 * p + 2] = colors[3 * v + 2];
        mask[p] = 1;
    }
}

extern "C" int pdhip_scatter_vertex_colors(const int32_t* texel, const float* colors, int V, float* atlas, uint8_t* mask,
                                           int32_t* owner_ws /*[A*A]*/, int A, void* stream) {
    PD_REQUIRE(texel && colors && atlas && mask && owner_ws && V > 0 && A > 0, "pdhip_scatter_vertex_colors: bad arguments");
    hipStream_t s = as_stream(stream);
    PD_HIP(hipMemsetAsync(owner_ws, 0xFF, (size_t)A * A * sizeof(int32_t), s));
    const int grid = min(cdiv(V, 256), 2048);
    k_vc_owner<<<grid, 256, 0, s>>>(texel, V, A, owner_ws);
    k_vc_write<<<grid, 256, 0, s>>>(texel, colors, V, A, owner_ws, atlas, mask);
    PD_LAUNCH_CHECK();
    return PDHIP_OK;
}
