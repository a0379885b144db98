/* TEST INFRASTRUCTURE ONLY — plain-C CPU restatement of the reference's hot path (see smpl_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. */
#ifndef SMPL_ORACLE_H
#define SMPL_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_JOINT_NUM 24
#define ORACLE_SHAPE_DIM 10
#define ORACLE_POSE_DIM 207
#define ORACLE_THETA_DIM 75 /* 3 * (JOINT_NUM + 1), node/node.cpp:787 */

typedef struct oracle_model oracle_model;

oracle_model * oracle_model_create(int64_t V, int64_t F, const float * vertices_template, const float * shape_blend_shapes,
                                   const float * pose_blend_shapes, const float * joint_regressor, const float * weights,
                                   const int64_t * kinematic_tree, const int32_t * face_indices_1based);
void oracle_model_destroy(oracle_model * m);
/* Override the per-vertex adjacent-face iteration order (the reference iterates an unordered_map, src/SMPL.cpp:529).
 * Default order is ascending face id. */
int oracle_model_set_adjacency(oracle_model * m, int64_t vertex, int64_t n, const int64_t * faces);
int64_t oracle_model_get_adjacency(const oracle_model * m, int64_t vertex, int64_t cap, int64_t * faces, float * w);

/* ---- stage-level restatements (for the Tester.cpp KATs) ---- */
void oracle_rodrigues(int64_t n, const float * theta /*[n,24,3]*/, float * rot /*[n,24,3,3]*/);
void oracle_blend_shape(int64_t V, int64_t n, const float * beta, const float * theta /*[n,24,3]*/, const float * S,
                        const float * P, float * shape_blend /*[n,V,3]*/, float * pose_blend /*[n,V,3]*/,
                        float * pose_rot /*[n,24,3,3]*/);
void oracle_joint_regression(int64_t V, int64_t n, const float * T, const float * Jreg, const float * shape_blend,
                             const float * pose_blend, float * rest /*[n,V,3]*/, float * joints /*[n,24,3]*/);
void oracle_world_transformation(int64_t n, const int64_t * kintree, const float * joints, const float * pose_rot,
                                 float * xforms /*[n,24,4,4]*/);
void oracle_lbs(int64_t V, int64_t n, const float * W, const float * rest, const float * xforms /*[n,24,4,4]*/,
                const float * root_pos /*[n,3] or NULL*/, float * verts /*[n,V,3]*/);

/* ---- SMPL::launch (src/SMPL.cpp:671-737); outputs may be NULL; threads <= 0 means all (OpenMP) ---- */
void oracle_fk(const oracle_model * m, int64_t n, const float * beta, const float * theta /*[n,25,3]*/, float * verts,
               float * joints, float * xforms, float * rest, float * pose_rot, int threads);
int oracle_max_threads(void);

/* ---- mesh queries ---- */
void oracle_face_normal(const oracle_model * m, const float * verts /*[V,3]*/, int64_t face, float * n3);
void oracle_vertex_normal(const oracle_model * m, const float * verts, int64_t vertex, float * n3);
void oracle_triangle_vertex_weights(const float * pos3, const float * tri9, float * w3);
/* igl::winding_number (node/node.cpp:1052) restated as the plain solid-angle sum, fp64: points [count,3] -> w [count] */
void oracle_winding_numbers(const oracle_model * m, const float * verts, int64_t count, const float * points, double * w);
/* igl::point_mesh_squared_distance restated as exact brute force (node/node.cpp:982) */
void oracle_closest_points(const oracle_model * m, const float * verts, int64_t K, const float * points /*[K,3]*/,
                           int64_t * face /*[K]*/, float * closest /*[K,3]*/, float * sqdist /*[K]*/);

/* ---- IK (node/node.cpp:704-1001), one frame, direct-theta layout ---- */
typedef struct
{
  int64_t K;
  int64_t * face_idx;        /* [K] 0-based, IkTask::faceIdx_ */
  float * vertex_weights;    /* [K,3] */
  float * tangents;          /* [K,3,2] */
  const float * target_pos;  /* [K,3] */
  const float * target_normal; /* [K,3] */
  const double * pos_task_weight;    /* [K] */
  const double * normal_task_weight; /* [K] */
  const double * phi_limit;          /* [K] */
  const double * normal_offset;      /* [K] */
} oracle_tasks;

/* node.cpp:798-877: e [4K], J [4K, 75 + 2K + (optimize_beta ? 10 : 0)] row-major; also refreshes tangents and
 * vertex weights (:803-804) and reports actualPos / actualNormal. verts_out (nullable) = FK vertices [V,3]. */
void oracle_ik_eval(const oracle_model * m, const float * beta, const float * theta, oracle_tasks * t, int optimize_beta,
                    float * actual_pos, float * actual_normal, double * e, double * J, float * verts_out);

/* node.cpp:883-904: A = J^T J + damping, b = J^T e. vposer_theta (nullable, [theta_dim]) adds the :895-904 prior. */
void oracle_normal_equations(int64_t rows, int64_t theta_dim, int64_t phi_dim, int64_t beta_dim, const double * e,
                             const double * J, const float * vposer_theta, double * A, double * b);
/* node.cpp:933-938: x = -LLT(A)^-1 b. Returns non-zero on a non-positive pivot. */
int oracle_llt_solve(int64_t D, const double * A, const double * b, double * x);
/* node.cpp:909-930: min 1/2 x'Ax + b'x s.t. lo <= x <= hi (QLD restated as a primal active-set box QP). */
int oracle_box_qp(int64_t D, const double * A, const double * b, const double * lo, const double * hi, double * x);

/* The whole loop body :704-1001 repeated `iters` times on one frame (enable_qp: use the box QP).
 * optimize_beta_from: iteration index from which beta is optimised and phi limits are live (-1 = never),
 * mirroring solveMocapBody (:655, :695).  theta/beta/tasks are updated in place. */
int oracle_ik_solve(const oracle_model * m, float * beta, float * theta, oracle_tasks * t, int iters, int enable_qp,
                    int optimize_beta_from, double * last_e_sqnorm);

#ifdef __cplusplus
}
#endif
#endif
