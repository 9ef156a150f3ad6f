/* Host-side helpers for the dataset layer (plain C, no GPU): CRC-32C for TFRecord framing and a fast parser
 * for Aftr frame text.  Reference call sites: tf.io.TFRecordWriter (pointcloud/PointCloudSet.py:251-288) and the
 * per-line float()/split loop of add_from_aftr_output (pointcloud/PointCloudSet.py:156-198). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static uint32_t crc_table[8][256];
static int crc_ready = 0;

static void crc_init(void) {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
    crc_table[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; ++i)
    for (int t = 1; t < 8; ++t) crc_table[t][i] = (crc_table[t - 1][i] >> 8) ^ crc_table[0][crc_table[t - 1][i] & 0xff];
  crc_ready = 1;
}

/* CRC-32C (Castagnoli), as used by TFRecord */
uint32_t pn_crc32c(const uint8_t* p, size_t n) {
  if (!crc_ready) crc_init();
  uint32_t c = 0xffffffffu;
  while (n >= 8) {
    uint32_t lo, hi;
    memcpy(&lo, p, 4);
    memcpy(&hi, p + 4, 4);
    lo ^= c;
    c = crc_table[7][lo & 0xff] ^ crc_table[6][(lo >> 8) & 0xff] ^ crc_table[5][(lo >> 16) & 0xff] ^ crc_table[4][lo >> 24] ^
        crc_table[3][hi & 0xff] ^ crc_table[2][(hi >> 8) & 0xff] ^ crc_table[1][(hi >> 16) & 0xff] ^ crc_table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = crc_table[0][(c ^ *p++) & 0xff] ^ (c >> 8);
  return c ^ 0xffffffffu;
}

uint32_t pn_masked_crc32c(const uint8_t* p, size_t n) {
  uint32_t c = pn_crc32c(p, n);
  return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

/* Parse one Aftr frame:  "(<x>, <y>, <z>) <class> <part>" per line (PointCloudSet.py:161-198).
 *   labels after ')' are the space-separated tokens longer than one character (:177); exactly two are required.
 *   class_names / part_names: arrays of NUL-terminated strings.
 * Outputs: xyz (max_pts*3 doubles), part (max_pts int32), *cls (last accepted class id), *non_finite (lines whose
 * coordinates are not finite and were skipped, :187-198).
 * Returns the number of accepted points, or a negative code:
 *   -1 malformed line / wrong number of labels, -2 unknown class label, -3 unknown part label, -4 more than max_pts.
 * On error *err_line holds the 0-based line number. */
static int lookup(const char* tok, size_t len, const char* const* names, int n) {
  for (int i = 0; i < n; ++i)
    if (strlen(names[i]) == len && memcmp(names[i], tok, len) == 0) return i;
  return -1;
}

long pn_parse_aftr_frame(const char* text, size_t n, const char* const* class_names, int n_class, const char* const* part_names,
                         int n_part, double* xyz, int32_t* part, long max_pts, int32_t* cls, long* non_finite, long* err_line) {
  long count = 0, line_no = 0;
  size_t i = 0;
  *non_finite = 0;
  *cls = -1;
  while (i < n) {
    size_t e = i;
    while (e < n && text[e] != '\n') ++e;
    /* strip */
    size_t a = i, b = e;
    while (a < b && (text[a] == ' ' || text[a] == '\t' || text[a] == '\r')) ++a;
    while (b > a && (text[b - 1] == ' ' || text[b - 1] == '\t' || text[b - 1] == '\r')) --b;
    if (b > a) {
      const char* l = text + a;
      size_t len = b - a;
      const char* po = memchr(l, '(', len);
      const char* pc = memchr(l, ')', len);
      if (!po || !pc || pc < po) { *err_line = line_no; return -1; }
      double v[3];
      int nv = 0;
      const char* s = po + 1;
      while (s < pc && nv < 3) {
        char* endp;
        char buf[64];
        const char* comma = memchr(s, ',', (size_t)(pc - s));
        size_t fl = (size_t)((comma ? comma : pc) - s);
        if (fl == 0 || fl >= sizeof(buf)) { *err_line = line_no; return -1; }
        memcpy(buf, s, fl);
        buf[fl] = 0;
        v[nv] = strtod(buf, &endp);
        while (*endp == ' ') ++endp;
        if (endp == buf || *endp != 0) { *err_line = line_no; return -1; }
        ++nv;
        s = comma ? comma + 1 : pc;
      }
      if (nv != 3 || s < pc) { *err_line = line_no; return -1; }
      /* labels */
      const char* t = pc + 1;
      const char* lend = l + len;
      const char* toks[4];
      size_t tl[4];
      int nt = 0;
      while (t < lend) {
        while (t < lend && *t == ' ') ++t;
        const char* ts = t;
        while (t < lend && *t != ' ') ++t;
        if (t - ts > 1) {
          if (nt < 4) { toks[nt] = ts; tl[nt] = (size_t)(t - ts); }
          ++nt;
        }
      }
      if (nt != 2) { *err_line = line_no; return -1; }
      int ci = lookup(toks[0], tl[0], class_names, n_class);
      if (ci < 0) { *err_line = line_no; return -2; }
      int pi = lookup(toks[1], tl[1], part_names, n_part);
      if (pi < 0) { *err_line = line_no; return -3; }
      if (isfinite(v[0]) && isfinite(v[1]) && isfinite(v[2])) {
        if (count >= max_pts) { *err_line = line_no; return -4; }
        xyz[3 * count] = v[0]; xyz[3 * count + 1] = v[1]; xyz[3 * count + 2] = v[2];
        part[count] = pi;
        *cls = ci;
        ++count;
      } else {
        ++*non_finite;
      }
    }
    i = e + 1;
    ++line_no;
  }
  return count;
}
