/* ORACLE (test infrastructure, not product code): rotated-box IoU and greedy NMS.
 *
 * The reference computes polygon IoU with a third-party dependency that is NOT vendored under
 * /root/reference: shapely (GEOS), unpinned in requirements.txt:14, called from
 * utils/common_utils.py:150-191 (convert_format / compute_iou) by utils/box_utils.py:823-868
 * (nms_rotated).  It is not installable here, so parity with GEOS itself is UNPINNED; this file
 * restates the published algorithm for the only case the path produces — two convex
 * quadrilaterals — as Sutherland-Hodgman clipping in double precision:
 *   iou = area(A ∩ B) / area(A ∪ B),   area(A ∪ B) = area(A) + area(B) - area(A ∩ B)
 * and follows nms_rotated's bookkeeping literally: order = argsort(scores)[::-1][:1000]
 * (ties: higher index first, i.e. a stable ascending sort reversed), then greedy removal of
 * iou > threshold with the iou array rounded to float32 before the comparison
 * (common_utils.py:174: np.array(iou, dtype=np.float32)).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

typedef struct { double x, y; } P2;

static double poly_area(const P2* p, int n) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) {
        const P2 *u = &p[i], *v = &p[(i + 1) % n];
        a += u->x * v->y - v->x * u->y;
    }
    return 0.5 * a;
}

/* clip polygon `in` (n vertices) against the half plane left of edge a->b (CCW clip polygon) */
static int clip_edge(const P2* in, int n, P2 a, P2 b, P2* out) {
    int m = 0;
    const double ex = b.x - a.x, ey = b.y - a.y;
    for (int i = 0; i < n; ++i) {
        const P2 cur = in[i], nxt = in[(i + 1) % n];
        const double dc = ex * (cur.y - a.y) - ey * (cur.x - a.x);
        const double dn = ex * (nxt.y - a.y) - ey * (nxt.x - a.x);
        if (dc >= 0.0) out[m++] = cur;
        if ((dc >= 0.0) != (dn >= 0.0)) {
            const double t = dc / (dc - dn);
            out[m].x = cur.x + t * (nxt.x - cur.x);
            out[m].y = cur.y + t * (nxt.y - cur.y);
            ++m;
        }
    }
    return m;
}

/* quads: 4 x (x,y) doubles each, any orientation */
double av2x_oracle_quad_iou(const double* qa, const double* qb) {
    P2 a[4], b[4], buf1[16], buf2[16];
    for (int i = 0; i < 4; ++i) { a[i].x = qa[2 * i]; a[i].y = qa[2 * i + 1]; b[i].x = qb[2 * i]; b[i].y = qb[2 * i + 1]; }
    double aa = poly_area(a, 4), ab = poly_area(b, 4);
    if (aa < 0) { P2 t = a[1]; a[1] = a[3]; a[3] = t; aa = -aa; }
    if (ab < 0) { P2 t = b[1]; b[1] = b[3]; b[3] = t; ab = -ab; }
    int n = 4;
    P2* src = buf1; P2* dst = buf2;
    for (int i = 0; i < 4; ++i) src[i] = a[i];
    for (int e = 0; e < 4 && n > 0; ++e) {
        n = clip_edge(src, n, b[e], b[(e + 1) % 4], dst);
        P2* t = src; src = dst; dst = t;
    }
    const double inter = n >= 3 ? fabs(poly_area(src, n)) : 0.0;
    const double uni = aa + ab - inter;
    return uni > 0.0 ? inter / uni : 0.0;
}

/* corners: (n, 4, 2) float32 (first four box corners, x/y); returns number kept, indices in keep */
int av2x_oracle_nms_rotated(const float* corners, const float* scores, int n, float threshold, int top, int32_t* keep) {
    if (n <= 0) return 0;
    int32_t* order = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    /* stable ascending by score (insertion into a merge-free O(n log n) would be nicer; n <= ~70k) */
    for (int i = 0; i < n; ++i) order[i] = i;
    /* simple stable merge sort */
    int32_t* tmp = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int i = lo, j = mid, k = lo;
            while (i < mid && j < hi) tmp[k++] = (scores[order[j]] < scores[order[i]]) ? order[j++] : order[i++];
            while (i < mid) tmp[k++] = order[i++];
            while (j < hi) tmp[k++] = order[j++];
        }
        int32_t* t = order; order = tmp; tmp = t;
    }
    int m = n < top ? n : top;
    int32_t* ixs = (int32_t*)malloc(sizeof(int32_t) * (size_t)m);
    for (int i = 0; i < m; ++i) ixs[i] = order[n - 1 - i];  /* [::-1][:top] */
    char* dead = (char*)calloc((size_t)m, 1);
    int nk = 0;
    for (int i = 0; i < m; ++i) {
        if (dead[i]) continue;
        keep[nk++] = ixs[i];
        double qa[8];
        for (int c = 0; c < 8; ++c) qa[c] = corners[(size_t)ixs[i] * 8 + c];
        for (int j = i + 1; j < m; ++j) {
            if (dead[j]) continue;
            double qb[8];
            for (int c = 0; c < 8; ++c) qb[c] = corners[(size_t)ixs[j] * 8 + c];
            const float iou = (float)av2x_oracle_quad_iou(qa, qb);
            if (iou > threshold) dead[j] = 1;
        }
    }
    free(order); free(tmp); free(ixs); free(dead);
    return nk;
}
