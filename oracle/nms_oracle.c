/*
 * oracle/nms_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Sequential restatement of the rotated BEV NMS the reference takes from mmdet3d v0.17.1
 * (pinned in /root/reference/README.md:153-156; NOT vendored in the reference tree):
 *   mmdet3d/ops/iou3d/src/iou3d_kernel.cu  (iou_bev, box_overlap, intersection, check_in_box2d,
 *   point_cmp, nms_kernel) and iou3d.cpp (the host loop over the suppression masks).
 * Reference call site: Anchor3DHead.get_bboxes -> box3d_multiclass_nms -> nms_gpu, configured by
 * projects/configs/bevfusion_NewScenes/bevfusion.py:147-155 (use_rotate_nms=True, nms_thr=0.2).
 *
 * PARITY UNPINNED: the reference repo holds no test, fixture or golden vector for this op and the
 * upstream package is absent from the image, so the algorithm (Sutherland-style polygon assembly:
 * 16 edge intersections + corner containment with a 1e-5 margin, bubble sort of the points by
 * atan2 around their mean, shoelace area; IoU = overlap / max(sa + sb - overlap, 1e-8); greedy
 * suppression in descending score order with `>` against the threshold) is restated from the
 * published source and anchored on the call site above plus hand-computed cases in
 * tests/test_oracle.py (axis-aligned overlaps, 45-degree squares, containment, disjoint boxes).
 *
 * Arithmetic contract shared with csrc/nms_rotated.hip so that IoUs agree bit for bit: float
 * operations in source order, no contraction (-ffp-contract=off), sin/cos/atan2 evaluated in double
 * and rounded to float.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  float x, y;
} point_t;

static const float EPS = 1e-8f;
static const float MARGIN = 1e-5f;

static float cross_o(point_t p1, point_t p2, point_t p0) {
  return (p1.x - p0.x) * (p2.y - p0.y) - (p2.x - p0.x) * (p1.y - p0.y);
}

static int check_rect_cross(point_t p1, point_t p2, point_t q1, point_t q2) {
  return fminf(p1.x, p2.x) <= fmaxf(q1.x, q2.x) && fminf(q1.x, q2.x) <= fmaxf(p1.x, p2.x) &&
         fminf(p1.y, p2.y) <= fmaxf(q1.y, q2.y) && fminf(q1.y, q2.y) <= fmaxf(p1.y, p2.y);
}

static int check_in_box2d(const float* box, point_t p) {
  float center_x = (box[0] + box[2]) / 2, center_y = (box[1] + box[3]) / 2;
  float angle_cos = (float)cos((double)(-box[4])), angle_sin = (float)sin((double)(-box[4]));
  float rot_x = (p.x - center_x) * angle_cos + (p.y - center_y) * angle_sin + center_x;
  float rot_y = -(p.x - center_x) * angle_sin + (p.y - center_y) * angle_cos + center_y;
  return rot_x > box[0] - MARGIN && rot_x < box[2] + MARGIN && rot_y > box[1] - MARGIN && rot_y < box[3] + MARGIN;
}

static int intersection(point_t p1, point_t p0, point_t q1, point_t q0, point_t* ans) {
  if (!check_rect_cross(p0, p1, q0, q1)) return 0;
  float s1 = cross_o(q0, p1, p0);
  float s2 = cross_o(p1, q1, p0);
  float s3 = cross_o(p0, q1, q0);
  float s4 = cross_o(q1, p1, q0);
  if (!(s1 * s2 > 0 && s3 * s4 > 0)) return 0;
  float s5 = cross_o(q1, p1, p0);
  if (fabsf(s5 - s1) > EPS) {
    ans->x = (s5 * q0.x - s1 * q1.x) / (s5 - s1);
    ans->y = (s5 * q0.y - s1 * q1.y) / (s5 - s1);
  } else {
    float a0 = p0.y - p1.y, b0 = p1.x - p0.x, c0 = p0.x * p1.y - p1.x * p0.y;
    float a1 = q0.y - q1.y, b1 = q1.x - q0.x, c1 = q0.x * q1.y - q1.x * q0.y;
    float D = a0 * b1 - a1 * b0;
    ans->x = (b0 * c1 - b1 * c0) / D;
    ans->y = (a1 * c0 - a0 * c1) / D;
  }
  return 1;
}

static void rotate_around_center(point_t center, float angle_cos, float angle_sin, point_t* p) {
  float new_x = (p->x - center.x) * angle_cos + (p->y - center.y) * angle_sin + center.x;
  float new_y = -(p->x - center.x) * angle_sin + (p->y - center.y) * angle_cos + center.y;
  p->x = new_x;
  p->y = new_y;
}

static int point_cmp(point_t a, point_t b, point_t center) {
  return (float)atan2((double)(a.y - center.y), (double)(a.x - center.x)) >
         (float)atan2((double)(b.y - center.y), (double)(b.x - center.x));
}

static float box_overlap(const float* box_a, const float* box_b) {
  float a_x1 = box_a[0], a_y1 = box_a[1], a_x2 = box_a[2], a_y2 = box_a[3], a_angle = box_a[4];
  float b_x1 = box_b[0], b_y1 = box_b[1], b_x2 = box_b[2], b_y2 = box_b[3], b_angle = box_b[4];
  point_t center_a = {(a_x1 + a_x2) / 2, (a_y1 + a_y2) / 2};
  point_t center_b = {(b_x1 + b_x2) / 2, (b_y1 + b_y2) / 2};
  point_t box_a_corners[5] = {{a_x1, a_y1}, {a_x2, a_y1}, {a_x2, a_y2}, {a_x1, a_y2}, {0, 0}};
  point_t box_b_corners[5] = {{b_x1, b_y1}, {b_x2, b_y1}, {b_x2, b_y2}, {b_x1, b_y2}, {0, 0}};
  float a_angle_cos = (float)cos((double)a_angle), a_angle_sin = (float)sin((double)a_angle);
  float b_angle_cos = (float)cos((double)b_angle), b_angle_sin = (float)sin((double)b_angle);
  for (int k = 0; k < 4; k++) {
    rotate_around_center(center_a, a_angle_cos, a_angle_sin, &box_a_corners[k]);
    rotate_around_center(center_b, b_angle_cos, b_angle_sin, &box_b_corners[k]);
  }
  box_a_corners[4] = box_a_corners[0];
  box_b_corners[4] = box_b_corners[0];

  point_t cross_points[16];
  point_t poly_center = {0, 0};
  int cnt = 0;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++)
      if (intersection(box_a_corners[i + 1], box_a_corners[i], box_b_corners[j + 1], box_b_corners[j],
                       &cross_points[cnt])) {
        poly_center.x = poly_center.x + cross_points[cnt].x;
        poly_center.y = poly_center.y + cross_points[cnt].y;
        cnt++;
      }
  for (int k = 0; k < 4; k++) {
    if (check_in_box2d(box_a, box_b_corners[k])) {
      poly_center.x = poly_center.x + box_b_corners[k].x;
      poly_center.y = poly_center.y + box_b_corners[k].y;
      cross_points[cnt++] = box_b_corners[k];
    }
    if (check_in_box2d(box_b, box_a_corners[k])) {
      poly_center.x = poly_center.x + box_a_corners[k].x;
      poly_center.y = poly_center.y + box_a_corners[k].y;
      cross_points[cnt++] = box_a_corners[k];
    }
  }
  if (cnt == 0) return 0.0f; /* upstream: 0/0 centre, no terms summed, area 0 */
  poly_center.x = poly_center.x / cnt;
  poly_center.y = poly_center.y / cnt;
  for (int j = 0; j < cnt - 1; j++)
    for (int i = 0; i < cnt - j - 1; i++)
      if (point_cmp(cross_points[i], cross_points[i + 1], poly_center)) {
        point_t t = cross_points[i];
        cross_points[i] = cross_points[i + 1];
        cross_points[i + 1] = t;
      }
  float area = 0;
  for (int k = 0; k < cnt - 1; k++) {
    point_t u = {cross_points[k].x - cross_points[0].x, cross_points[k].y - cross_points[0].y};
    point_t v = {cross_points[k + 1].x - cross_points[0].x, cross_points[k + 1].y - cross_points[0].y};
    area = area + (u.x * v.y - u.y * v.x);
  }
  return fabsf(area) / 2.0f;
}

static float iou_bev(const float* box_a, const float* box_b) {
  float sa = (box_a[2] - box_a[0]) * (box_a[3] - box_a[1]);
  float sb = (box_b[2] - box_b[0]) * (box_b[3] - box_b[1]);
  float s_overlap = box_overlap(box_a, box_b);
  return s_overlap / fmaxf(sa + sb - s_overlap, EPS);
}

/* out[i*nb+j] = IoU(a_i, b_j);  boxes are (x1,y1,x2,y2,angle). */
void oracle_iou_bev_matrix(const float* a, int na, const float* b, int nb, float* out) {
  for (int i = 0; i < na; ++i)
    for (int j = 0; j < nb; ++j) out[(size_t)i * nb + j] = iou_bev(a + (size_t)i * 5, b + (size_t)j * 5);
}

/* boxes already in descending score order; keep[] receives the surviving positions; returns count.
 * Same decisions as the 64-bit mask kernel + host loop: box j is suppressed by the first kept i<j
 * with IoU(box_i, box_j) > thresh (row box first, column box second, as in nms_kernel). */
int oracle_nms_rotated(const float* boxes, int n, float thresh, long long* keep) {
  unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int kept = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[kept++] = i;
    for (int j = i + 1; j < n; ++j)
      if (!removed[j] && iou_bev(boxes + (size_t)i * 5, boxes + (size_t)j * 5) > thresh) removed[j] = 1;
  }
  free(removed);
  return kept;
}
