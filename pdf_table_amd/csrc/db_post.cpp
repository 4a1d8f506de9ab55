// db_post.cpp -- host half of DBPostProcess.boxes_from_bitmap (reference:
// src/pdftable/model/db_pp/processor_ocr_db_pp.py:174-251; torch flavour db_net/ocr_detection_utils.py:167-256).
//
// The reference delegates this arithmetic to OpenCV (findContours, minAreaRect, boxPoints), pyclipper
// (PyclipperOffset, JT_ROUND) and shapely (area, length) -- none of which is vendored in the reference or
// present in the build image.  What follows restates their published algorithms:
//   * Suzuki-Abe border following as implemented by OpenCV's icvFetchContour (8-connected, RETR_LIST,
//     CHAIN_APPROX_SIMPLE), contours returned in reverse discovery order;
//   * convex hull (monotone chain) + rotating calipers (OpenCV rotatingCalipers, CALIPERS_MINAREARECT)
//     + RotatedRect::points;
//   * Angus Johnson's ClipperOffset for one closed path with round joins (arc tolerance 0.25).
// Pure CPU, no HIP calls: it sits either side of the device box-score kernel.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <thread>
#include <vector>

#include "../../include/pdftable_hip.h"

void pt_set_error(const char* fmt, ...);

namespace {

struct Pt {
  int x, y;
};
struct Pf {
  float x, y;
};

// ---- contour tracing ---------------------------------------------------------------------------------
// img: (h+2) x (w+2) int8 with a zero frame, 0/1 inside.  Marks follow OpenCV: 2 = visited border pixel,
// -126 (2 | -128) = visited border pixel whose right neighbour was examined and is background.
static const int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

static void fetch_contour(int8_t* img, int step, int ox, int oy, bool is_hole, std::vector<Pt>& out) {
  int deltas[16];
  for (int i = 0; i < 8; ++i) deltas[i] = deltas[i + 8] = kDy[i] * step + kDx[i];
  int8_t* i0 = img + oy * step + ox;
  int8_t* i1 = i0;
  int8_t* i3;
  int8_t* i4 = nullptr;
  int s_end, s;
  s_end = s = is_hole ? 0 : 4;
  Pt pt = {ox, oy};
  do {
    s = (s - 1) & 7;
    i1 = i0 + deltas[s];
    if (*i1 != 0) break;
  } while (s != s_end);
  if (s == s_end) {  // isolated pixel
    *i0 = (int8_t)(2 | -128);
    out.push_back(pt);
    return;
  }
  i3 = i0;
  int prev_s = s ^ 4;
  for (;;) {
    s_end = s;
    for (;;) {
      i4 = i3 + deltas[++s];
      if (*i4 != 0) break;
    }
    s &= 7;
    if ((unsigned)(s - 1) < (unsigned)s_end) {
      *i3 = (int8_t)(2 | -128);
    } else if (*i3 == 1) {
      *i3 = 2;
    }
    if (s != prev_s) {  // CHAIN_APPROX_SIMPLE: keep a point only where the direction changes
      out.push_back(pt);
      prev_s = s;
    }
    pt.x += kDx[s];
    pt.y += kDy[s];
    if (i4 == i0 && i3 == i1) break;
    i3 = i4;
    s = (s + 4) & 7;
  }
}

static void find_contours(const uint32_t* bitmap, int h, int w, std::vector<std::vector<Pt>>& contours) {
  const int step = w + 2;
  std::vector<int8_t> buf((size_t)(h + 2) * step, 0);
  const int wpr = w / 32;
  for (int y = 0; y < h; ++y) {
    int8_t* row = buf.data() + (size_t)(y + 1) * step + 1;
    for (int wx = 0; wx < wpr; ++wx) {
      uint32_t bits = bitmap[(size_t)y * wpr + wx];
      while (bits) {
        const int b = __builtin_ctz(bits);
        row[wx * 32 + b] = 1;
        bits &= bits - 1;
      }
    }
  }
  int8_t* img = buf.data();
  for (int y = 1; y <= h; ++y) {
    int8_t* row = img + (size_t)y * step;
    int prev = 0;
    for (int x = 1; x <= w; ++x) {  // OpenCV scans columns 1 .. padded_width-2
      const int p = row[x];
      if (p != prev) {
        bool is_hole = false;
        bool start = false;
        if (prev == 0 && p == 1) {
          start = true;
        } else if (p == 0 && prev >= 1) {
          start = true;
          is_hole = true;
        }
        if (start) {
          std::vector<Pt> c;
          fetch_contour(img, step, x - (is_hole ? 1 : 0), y, is_hole, c);
          for (auto& q : c) { q.x -= 1; q.y -= 1; }
          contours.push_back(std::move(c));
          // the origin pixel may have been re-marked by the trace
        }
        prev = row[x];
      }
    }
  }
  std::reverse(contours.begin(), contours.end());  // cvInsertNodeIntoTree prepends: last found comes first
}

// ---- min-area rectangle ---------------------------------------------------------------------------------
static double cross(const Pf& o, const Pf& a, const Pf& b) {
  return ((double)a.x - o.x) * ((double)b.y - o.y) - ((double)a.y - o.y) * ((double)b.x - o.x);
}

// strictly convex hull, counter-clockwise in a y-up frame, starting at the lowest-x (then lowest-y) point
static void convex_hull(std::vector<Pf> pts, std::vector<Pf>& hull) {
  std::sort(pts.begin(), pts.end(), [](const Pf& a, const Pf& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
  pts.erase(std::unique(pts.begin(), pts.end(), [](const Pf& a, const Pf& b) { return a.x == b.x && a.y == b.y; }),
            pts.end());
  const int n = (int)pts.size();
  hull.clear();
  if (n <= 2) {
    hull = pts;
    return;
  }
  std::vector<Pf> H(2 * n);
  int k = 0;
  for (int i = 0; i < n; ++i) {
    while (k >= 2 && cross(H[k - 2], H[k - 1], pts[i]) <= 0) --k;
    H[k++] = pts[i];
  }
  for (int i = n - 2, t = k + 1; i >= 0; --i) {
    while (k >= t && cross(H[k - 2], H[k - 1], pts[i]) <= 0) --k;
    H[k++] = pts[i];
  }
  H.resize(k - 1);
  hull = H;
}

struct RRect {
  float cx, cy, w, h, angle;
};

static void rotating_calipers(const Pf* points, int n, float out[6]) {
  float minarea = 3.402823466e+38f;
  std::vector<float> inv_len(n);
  std::vector<Pf> vect(n);
  int left = 0, bottom = 0, right = 0, top = 0;
  int seq[4] = {-1, -1, -1, -1};
  float orientation = 0;
  float base_a, base_b = 0;
  float left_x, right_x, top_y, bottom_y;
  Pf pt0 = points[0];
  left_x = right_x = pt0.x;
  top_y = bottom_y = pt0.y;
  for (int i = 0; i < n; i++) {
    if (pt0.x < left_x) left_x = pt0.x, left = i;
    if (pt0.x > right_x) right_x = pt0.x, right = i;
    if (pt0.y > top_y) top_y = pt0.y, top = i;
    if (pt0.y < bottom_y) bottom_y = pt0.y, bottom = i;
    const Pf pt = points[(i + 1 < n) ? i + 1 : 0];
    const double dx = (double)pt.x - pt0.x, dy = (double)pt.y - pt0.y;
    vect[i].x = (float)dx;
    vect[i].y = (float)dy;
    inv_len[i] = (float)(1. / sqrt(dx * dx + dy * dy));
    pt0 = pt;
  }
  {
    double ax = vect[n - 1].x, ay = vect[n - 1].y;
    for (int i = 0; i < n; i++) {
      const double bx = vect[i].x, by = vect[i].y;
      const double convexity = ax * by - ay * bx;
      if (convexity != 0) {
        orientation = (convexity > 0) ? 1.f : -1.f;
        break;
      }
      ax = bx;
      ay = by;
    }
    if (orientation == 0) orientation = 1.f;
  }
  base_a = orientation;
  seq[0] = bottom; seq[1] = right; seq[2] = top; seq[3] = left;
  int best_left = 0, best_bottom = 0;
  float bA = 1, bB = 0, bW = 0, bH = 0;
  for (int k = 0; k < n; k++) {
    const float dp[4] = {
        +base_a * vect[seq[0]].x + base_b * vect[seq[0]].y,
        -base_b * vect[seq[1]].x + base_a * vect[seq[1]].y,
        -base_a * vect[seq[2]].x - base_b * vect[seq[2]].y,
        +base_b * vect[seq[3]].x - base_a * vect[seq[3]].y,
    };
    float maxcos = dp[0] * inv_len[seq[0]];
    int main_element = 0;
    for (int i = 1; i < 4; ++i) {
      const float cosalpha = dp[i] * inv_len[seq[i]];
      if (cosalpha > maxcos) {
        main_element = i;
        maxcos = cosalpha;
      }
    }
    {
      const int pindex = seq[main_element];
      const float lead_x = vect[pindex].x * inv_len[pindex];
      const float lead_y = vect[pindex].y * inv_len[pindex];
      switch (main_element) {
        case 0: base_a = lead_x; base_b = lead_y; break;
        case 1: base_a = lead_y; base_b = -lead_x; break;
        case 2: base_a = -lead_x; base_b = -lead_y; break;
        default: base_a = -lead_y; base_b = lead_x; break;
      }
    }
    seq[main_element] += 1;
    seq[main_element] = (seq[main_element] == n) ? 0 : seq[main_element];
    float dx = points[seq[1]].x - points[seq[3]].x;
    float dy = points[seq[1]].y - points[seq[3]].y;
    const float width = dx * base_a + dy * base_b;
    dx = points[seq[2]].x - points[seq[0]].x;
    dy = points[seq[2]].y - points[seq[0]].y;
    const float height = -dx * base_b + dy * base_a;
    const float area = width * height;
    if (area <= minarea) {
      minarea = area;
      best_left = seq[3];
      bA = base_a; bW = width; bB = base_b; bH = height;
      best_bottom = seq[0];
    }
  }
  const float A1 = bA, B1 = bB, A2 = -bB, B2 = bA;
  const float C1 = A1 * points[best_left].x + points[best_left].y * B1;
  const float C2 = A2 * points[best_bottom].x + points[best_bottom].y * B2;
  const float idet = 1.f / (A1 * B2 - A2 * B1);
  out[0] = (C1 * B2 - C2 * B1) * idet;
  out[1] = (A1 * C2 - A2 * C1) * idet;
  out[2] = A1 * bW; out[3] = B1 * bW;
  out[4] = A2 * bH; out[5] = B2 * bH;
}

static RRect min_area_rect(const std::vector<Pf>& pts) {
  std::vector<Pf> hull;
  convex_hull(pts, hull);
  RRect box = {0, 0, 0, 0, 0};
  const int n = (int)hull.size();
  if (n > 2) {
    float out[6];
    rotating_calipers(hull.data(), n, out);
    box.cx = out[0] + (out[2] + out[4]) * 0.5f;
    box.cy = out[1] + (out[3] + out[5]) * 0.5f;
    box.w = (float)sqrt((double)out[2] * out[2] + (double)out[3] * out[3]);
    box.h = (float)sqrt((double)out[4] * out[4] + (double)out[5] * out[5]);
    box.angle = (float)atan2((double)out[3], (double)out[2]);
  } else if (n == 2) {
    box.cx = (hull[0].x + hull[1].x) * 0.5f;
    box.cy = (hull[0].y + hull[1].y) * 0.5f;
    const double dx = (double)hull[1].x - hull[0].x, dy = (double)hull[1].y - hull[0].y;
    box.w = (float)sqrt(dx * dx + dy * dy);
    box.h = 0;
    box.angle = (float)atan2(dy, dx);
  } else if (n == 1) {
    box.cx = hull[0].x;
    box.cy = hull[0].y;
  }
  box.angle = (float)(box.angle * 180 / 3.1415926535897932384626433832795);
  return box;
}

static void box_points(const RRect& r, Pf pt[4]) {
  const double ang = r.angle * 3.1415926535897932384626433832795 / 180.;
  const float b = (float)cos(ang) * 0.5f;
  const float a = (float)sin(ang) * 0.5f;
  pt[0].x = r.cx - a * r.h - b * r.w;
  pt[0].y = r.cy + b * r.h - a * r.w;
  pt[1].x = r.cx + a * r.h - b * r.w;
  pt[1].y = r.cy - b * r.h - a * r.w;
  pt[2].x = 2 * r.cx - pt[0].x;
  pt[2].y = 2 * r.cy - pt[0].y;
  pt[3].x = 2 * r.cx - pt[1].x;
  pt[3].y = 2 * r.cy - pt[1].y;
}

// get_mini_boxes (processor_ocr_db_pp.py:230-251): stable sort by x, then the y rule -> TL, TR, BR, BL
static float mini_box(const std::vector<Pf>& pts, Pf box[4]) {
  const RRect r = min_area_rect(pts);
  Pf p[4];
  box_points(r, p);
  std::stable_sort(p, p + 4, [](const Pf& a, const Pf& b) { return a.x < b.x; });
  int i1, i2, i3, i4;
  if (p[1].y > p[0].y) { i1 = 0; i4 = 1; } else { i1 = 1; i4 = 0; }
  if (p[3].y > p[2].y) { i2 = 2; i3 = 3; } else { i2 = 3; i3 = 2; }
  box[0] = p[i1]; box[1] = p[i2]; box[2] = p[i3]; box[3] = p[i4];
  return r.w < r.h ? r.w : r.h;
}

// ---- ClipperOffset, one closed path, round joins ---------------------------------------------------------
typedef long long cInt;
static inline cInt cround(double v) { return (v < 0) ? (cInt)(v - 0.5) : (cInt)(v + 0.5); }

static void clipper_offset_round(const cInt sx[], const cInt sy[], int len_in, double delta, std::vector<Pf>& out) {
  // AddPath: strip consecutive duplicates (closed path: also last == first)
  std::vector<cInt> X, Y;
  for (int i = 0; i < len_in; ++i) {
    if (!X.empty() && X.back() == sx[i] && Y.back() == sy[i]) continue;
    X.push_back(sx[i]);
    Y.push_back(sy[i]);
  }
  while (X.size() > 1 && X.back() == X[0] && Y.back() == Y[0]) { X.pop_back(); Y.pop_back(); }
  const int len = (int)X.size();
  out.clear();
  if (len < 3) return;
  // FixOrientations: Area >= 0 is the accepted orientation
  double a2 = 0;
  for (int i = 0, j = len - 1; i < len; ++i) {
    a2 += ((double)X[j] + (double)X[i]) * ((double)Y[j] - (double)Y[i]);
    j = i;
  }
  if (-a2 * 0.5 < 0) {
    std::reverse(X.begin(), X.end());
    std::reverse(Y.begin(), Y.end());
  }
  const double pi = 3.141592653589793238;
  const double two_pi = pi * 2;
  const double def_arc_tolerance = 0.25;
  const double arc_tol = 0.25;  // pyclipper.PyclipperOffset default
  double y;
  if (arc_tol > fabs(delta) * def_arc_tolerance) y = fabs(delta) * def_arc_tolerance; else y = arc_tol;
  double steps = pi / acos(1 - y / fabs(delta));
  if (steps > fabs(delta) * pi) steps = fabs(delta) * pi;
  double m_sin = sin(two_pi / steps);
  const double m_cos = cos(two_pi / steps);
  const double steps_per_rad = steps / two_pi;
  if (delta < 0.0) m_sin = -m_sin;
  std::vector<double> nx(len), ny(len);
  for (int j = 0; j < len; ++j) {
    const int j2 = (j + 1 == len) ? 0 : j + 1;
    double dx = (double)(X[j2] - X[j]), dy = (double)(Y[j2] - Y[j]);
    if (dx == 0 && dy == 0) { nx[j] = ny[j] = 0; continue; }
    const double f = 1.0 / sqrt(dx * dx + dy * dy);
    dx *= f; dy *= f;
    nx[j] = dy; ny[j] = -dx;
  }
  auto push = [&](cInt px, cInt py) { out.push_back(Pf{(float)px, (float)py}); };
  int k = len - 1;
  for (int j = 0; j < len; ++j) {
    double sinA = nx[k] * ny[j] - nx[j] * ny[k];
    bool done = false;
    if (fabs(sinA * delta) < 1.0) {
      const double cosA = nx[k] * nx[j] + ny[j] * ny[k];
      if (cosA > 0) {
        push(cround(X[j] + nx[k] * delta), cround(Y[j] + ny[k] * delta));
        done = true;
      }
    } else if (sinA > 1.0) sinA = 1.0; else if (sinA < -1.0) sinA = -1.0;
    if (!done) {
      if (sinA * delta < 0) {
        push(cround(X[j] + nx[k] * delta), cround(Y[j] + ny[k] * delta));
        push(X[j], Y[j]);
        push(cround(X[j] + nx[j] * delta), cround(Y[j] + ny[j] * delta));
      } else {
        const double a = atan2(sinA, nx[k] * nx[j] + ny[k] * ny[j]);
        int st = (int)cround(steps_per_rad * fabs(a));
        if (st < 1) st = 1;
        double Xn = nx[k], Yn = ny[k], X2;
        for (int i = 0; i < st; ++i) {
          push(cround(X[j] + Xn * delta), cround(Y[j] + Yn * delta));
          X2 = Xn;
          Xn = Xn * m_cos - m_sin * Yn;
          Yn = X2 * m_sin + Yn * m_cos;
        }
        push(cround(X[j] + nx[j] * delta), cround(Y[j] + ny[j] * delta));
      }
    }
    k = j;
  }
}

}  // namespace

extern "C" {

int pt_db_candidates(const uint32_t* h_bitmap, int net_h, int net_w, int max_candidates, float min_size, float* h_boxes,
                     float* h_sside, int cap, int* n_out) {
  if (!h_bitmap || !h_boxes || !n_out || net_h <= 0 || net_w <= 0 || net_w % 32 != 0) {
    pt_set_error("pt_db_candidates: bad arguments");
    return PT_ERR_INVALID;
  }
  std::vector<std::vector<Pt>> contours;
  find_contours(h_bitmap, net_h, net_w, contours);
  const int nc = std::min((int)contours.size(), max_candidates);
  int n = 0;
  std::vector<Pf> pts;
  for (int i = 0; i < nc && n < cap; ++i) {
    pts.clear();
    for (const Pt& q : contours[i]) pts.push_back(Pf{(float)q.x, (float)q.y});
    Pf box[4];
    const float sside = mini_box(pts, box);
    if (sside < min_size) continue;
    for (int k = 0; k < 4; ++k) {
      h_boxes[(size_t)n * 8 + 2 * k] = box[k].x;
      h_boxes[(size_t)n * 8 + 2 * k + 1] = box[k].y;
    }
    if (h_sside) h_sside[n] = sside;
    ++n;
  }
  *n_out = n;
  return PT_OK;
}

int pt_db_finalize(const float* h_boxes, const float* h_scores, int nb, float box_thresh, float unclip_ratio,
                   float min_size, int net_h, int net_w, int dest_h, int dest_w, int post_flavour, int32_t* h_out,
                   float* h_out_scores, int cap, int* n_out) {
  if ((nb > 0 && (!h_boxes || !h_scores)) || !h_out || !n_out) {
    pt_set_error("pt_db_finalize: bad arguments");
    return PT_ERR_INVALID;
  }
  int n = 0;
  std::vector<Pf> off;
  for (int i = 0; i < nb && n < cap; ++i) {
    const float score = h_scores[i];
    if (box_thresh > score) continue;
    const float* b = h_boxes + (size_t)i * 8;
    // unclip: shapely area / length on the float quad, then Clipper on truncated integer vertices
    double area2 = 0, perim = 0;
    for (int k = 0; k < 4; ++k) {
      const int k2 = (k + 1) & 3;
      const double x0 = b[2 * k], y0 = b[2 * k + 1], x1 = b[2 * k2], y1 = b[2 * k2 + 1];
      area2 += x0 * y1 - x1 * y0;
      perim += sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0));
    }
    const double area = fabs(area2) * 0.5;
    if (perim <= 0) continue;
    const double distance = area * (double)unclip_ratio / perim;
    cInt sx[4], sy[4];
    for (int k = 0; k < 4; ++k) {
      sx[k] = (cInt)b[2 * k];  // Cython <cInt> cast: truncation toward zero
      sy[k] = (cInt)b[2 * k + 1];
    }
    clipper_offset_round(sx, sy, 4, distance, off);
    if (off.empty()) continue;
    Pf box[4];
    const float sside = mini_box(off, box);
    if (sside < min_size + 2) continue;
    for (int k = 0; k < 4; ++k) {
      // np.clip(np.round(box / width * dest_width), 0, dest_width): float32 divide, then double (numpy>=2
      // promotes float32 array * np.float64 scalar to float64), round half to even, astype(int16)
      double vx, vy;
      if (post_flavour == PT_DET_POST_DB_TORCH) {
        // np.array(box).astype(np.int32) first; int / int is float64 in numpy
        const double ix = (double)(int32_t)box[k].x, iy = (double)(int32_t)box[k].y;
        vx = nearbyint(ix / (double)net_w * (double)dest_w);
        vy = nearbyint(iy / (double)net_h * (double)dest_h);
      } else {
        const float qx = box[k].x / (float)net_w, qy = box[k].y / (float)net_h;
        vx = nearbyint((double)qx * (double)dest_w);
        vy = nearbyint((double)qy * (double)dest_h);
      }
      vx = vx < 0 ? 0 : (vx > dest_w ? dest_w : vx);
      vy = vy < 0 ? 0 : (vy > dest_h ? dest_h : vy);
      if (post_flavour == PT_DET_POST_DB_TORCH) {
        h_out[(size_t)n * 8 + 2 * k] = (int32_t)(long long)vx;
        h_out[(size_t)n * 8 + 2 * k + 1] = (int32_t)(long long)vy;
      } else {
        h_out[(size_t)n * 8 + 2 * k] = (int32_t)(int16_t)(long long)vx;
        h_out[(size_t)n * 8 + 2 * k + 1] = (int32_t)(int16_t)(long long)vy;
      }
    }
    if (h_out_scores) h_out_scores[n] = score;
    ++n;
  }
  *n_out = n;
  return PT_OK;
}

// ---- batch forms: all pages of a batch in one call, spread over threads inside the library --------------------------
// The per-page calls above are what DBPostProcess does for one image; a page batch used to go through a Python thread
// pool that called them page by page (two pool.map rounds, per-page numpy glue under the GIL): 25 ms per 64 pages with ~80
// boxes each, more than the GPU needs for the network.  Here one call walks the pages with an atomic counter.
}  // extern "C"

template <typename F>
static void parallel_pages(int n, int n_threads, F f) {
  int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
  if (nt > n) nt = n;
  if (nt <= 1) {
    for (int i = 0; i < n; ++i) f(i);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  th.reserve(nt);
  for (int t = 0; t < nt; ++t)
    th.emplace_back([&]() {
      for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
    });
  for (auto& t : th) t.join();
}

static int filter_tag(const int32_t* in, int n, int img_h, int img_w, float* out);

extern "C" {

int pt_db_candidates_batch(const uint32_t* h_bitmaps, int n, int net_h, int net_w, int max_candidates, float min_size,
                           int n_threads, float* h_boxes, float* h_sside, int cap, int* n_out) {
  if (!h_bitmaps || !h_boxes || !n_out || n < 0 || cap <= 0 || net_h <= 0 || net_w <= 0 || net_w % 32 != 0) {
    pt_set_error("pt_db_candidates_batch: bad arguments");
    return PT_ERR_INVALID;
  }
  const size_t words = (size_t)net_h * (net_w / 32);
  std::atomic<int> bad(0);
  parallel_pages(n, n_threads, [&](int i) {
    const int rc = pt_db_candidates(h_bitmaps + (size_t)i * words, net_h, net_w, max_candidates, min_size,
                                    h_boxes + (size_t)i * cap * 8, h_sside ? h_sside + (size_t)i * cap : nullptr, cap, n_out + i);
    if (rc != PT_OK) bad.store(rc);
  });
  return bad.load();
}

}  // extern "C"

// filter_tag_det_res of PPOcrDetectionPostProcessor (db_pp/processor_ocr_db_pp.py:344-386) on one page's int boxes, in
// place: order_points_clockwise (sort by x -- numpy's argsort on 4 elements is an insertion sort, i.e. stable --, the two
// left points and the two right points by y), clip to the page, drop boxes whose int(norm) width or height is <= 3.
// boxes come in as int32 [n][8] and leave as float32 [n'][8] (the reference returns float32 points).
static int filter_tag(const int32_t* in, int n, int img_h, int img_w, float* out) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    float px[4], py[4];
    for (int k = 0; k < 4; ++k) {
      px[k] = (float)in[(size_t)i * 8 + 2 * k];
      py[k] = (float)in[(size_t)i * 8 + 2 * k + 1];
    }
    int o[4] = {0, 1, 2, 3};
    std::stable_sort(o, o + 4, [&](int a, int b) { return px[a] < px[b]; });
    int l0 = o[0], l1 = o[1], r0 = o[2], r1 = o[3];
    if (py[l1] < py[l0]) std::swap(l0, l1);        // stable: equal y keeps the x order
    if (py[r1] < py[r0]) std::swap(r0, r1);
    const int ord[4] = {l0, r0, r1, l1};            // tl, tr, br, bl
    float rx[4], ry[4];
    for (int k = 0; k < 4; ++k) {
      float x = px[ord[k]], y = py[ord[k]];
      x = x < 0.f ? 0.f : (x > (float)(img_w - 1) ? (float)(img_w - 1) : x);
      y = y < 0.f ? 0.f : (y > (float)(img_h - 1) ? (float)(img_h - 1) : y);
      rx[k] = truncf(x);
      ry[k] = truncf(y);
    }
    const float wx = rx[0] - rx[1], wy = ry[0] - ry[1], hx = rx[0] - rx[3], hy = ry[0] - ry[3];
    const long long w = (long long)sqrtf(wx * wx + wy * wy), h = (long long)sqrtf(hx * hx + hy * hy);
    if (w <= 3 || h <= 3) continue;
    for (int k = 0; k < 4; ++k) {
      out[(size_t)m * 8 + 2 * k] = rx[k];
      out[(size_t)m * 8 + 2 * k + 1] = ry[k];
    }
    ++m;
  }
  return m;
}

extern "C" {

int pt_db_finalize_batch(const float* h_boxes, const float* h_scores, const int* nb, int n, int cap, float box_thresh,
                         float unclip_ratio, float min_size, int net_h, int net_w, int dest_h, int dest_w, int post_flavour,
                         int filter, int n_threads, int32_t* h_out, float* h_out_f32, float* h_out_scores, int* n_out) {
  if (!h_boxes || !h_scores || !nb || !h_out || !n_out || n < 0 || cap <= 0 || (filter && !h_out_f32)) {
    pt_set_error("pt_db_finalize_batch: bad arguments");
    return PT_ERR_INVALID;
  }
  std::atomic<int> bad(0);
  parallel_pages(n, n_threads, [&](int i) {
    int k = 0;
    const int rc = pt_db_finalize(h_boxes + (size_t)i * cap * 8, h_scores + (size_t)i * cap, nb[i], box_thresh, unclip_ratio, min_size,
                                  net_h, net_w, dest_h, dest_w, post_flavour, h_out + (size_t)i * cap * 8,
                                  h_out_scores ? h_out_scores + (size_t)i * cap : nullptr, cap, &k);
    if (rc != PT_OK) { bad.store(rc); n_out[i] = 0; return; }
    n_out[i] = filter ? filter_tag(h_out + (size_t)i * cap * 8, k, dest_h, dest_w, h_out_f32 + (size_t)i * cap * 8) : k;
  });
  return bad.load();
}

// ---- layout: greedy hard NMS of OCRPicodetPostProcessor (picodet/processor_picodet.py:301-348), host side ----------
// One call per page: n_groups independent candidate lists (one per class).  boxes: float64 [n][5] (x1, y1, x2, y2, score);
// order: for every group the candidate indices in ASCENDING score order as numpy's argsort()[-candidate_size:] gave them
// (the tie order is numpy's, so it is computed there); group_off: [n_groups + 1] offsets into `order`.
// picked: [sum of group sizes] indices in pick order, per group; n_picked: [n_groups].
// iou_of in float64, same operation order: inter / ((a0 + a1) - inter + eps).
int pt_hard_nms(const double* boxes, const int64_t* order, const int64_t* group_off, int n_groups, double iou_threshold,
                int top_k, int64_t* picked, int32_t* n_picked) {
  if (!boxes || !order || !group_off || !picked || !n_picked || n_groups < 0) {
    pt_set_error("pt_hard_nms: bad arguments");
    return PT_ERR_INVALID;
  }
  const double eps = 1e-5;
  std::vector<int64_t> idx;
  for (int g = 0; g < n_groups; ++g) {
    idx.assign(order + group_off[g], order + group_off[g + 1]);
    int64_t* out = picked + group_off[g];
    int np = 0;
    while (!idx.empty()) {
      const int64_t cur = idx.back();
      out[np++] = cur;
      if ((top_k > 0 && np == top_k) || idx.size() == 1) break;
      idx.pop_back();
      const double* c = boxes + cur * 5;
      double cw = c[2] - c[0], ch = c[3] - c[1];
      if (cw < 0.0) cw = 0.0;
      if (ch < 0.0) ch = 0.0;
      const double a1 = cw * ch;
      size_t keep = 0;
      for (size_t k = 0; k < idx.size(); ++k) {
        const double* b = boxes + idx[k] * 5;
        double w = (b[2] < c[2] ? b[2] : c[2]) - (b[0] > c[0] ? b[0] : c[0]);
        double h = (b[3] < c[3] ? b[3] : c[3]) - (b[1] > c[1] ? b[1] : c[1]);
        if (w < 0.0) w = 0.0;
        if (h < 0.0) h = 0.0;
        const double inter = w * h;
        double bw = b[2] - b[0], bh = b[3] - b[1];
        if (bw < 0.0) bw = 0.0;
        if (bh < 0.0) bh = 0.0;
        const double iou = inter / (bw * bh + a1 - inter + eps);
        if (iou <= iou_threshold) idx[keep++] = idx[k];
      }
      idx.resize(keep);
    }
    n_picked[g] = np;
  }
  return PT_OK;
}

}  // extern "C"
