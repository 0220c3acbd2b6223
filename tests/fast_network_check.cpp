// Host check of csrc/orb.hip's FAST-9 score (fast_score_raw: raw circle values, compass pre-test by the second largest / second
// smallest of four, ONE polarity per lane, van Herk arc network) against the two-sided doubling network on biased values it
// replaced in round 4: the same arithmetic in plain C++ (16-bit unsigned minima / maxima), random and adversarial circles.
//   g++ -O2 -o /tmp/fast_network_check tests/fast_network_check.cpp && /tmp/fast_network_check [cases per mode]
// tests/test_fast_network_host.py runs it in the CPU suite.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
static inline uint32_t min16(uint32_t a, uint32_t b) { return std::min<uint16_t>(a, b); }
static inline uint32_t max16(uint32_t a, uint32_t b) { return std::max<uint16_t>(a, b); }
static int old_score(const uint32_t (&e)[16], int thr) {
  uint32_t lo2[16], hi2[16], lo4[16], hi4[16];
  for (int k = 0; k < 16; ++k) { lo2[k] = min16(e[k], e[(k + 1) & 15]); hi2[k] = max16(e[k], e[(k + 1) & 15]); }
  for (int k = 0; k < 16; ++k) { lo4[k] = min16(lo2[k], lo2[(k + 2) & 15]); hi4[k] = max16(hi2[k], hi2[(k + 2) & 15]); }
  uint32_t A = 0u, B = 0xFFFFu;
  for (int k = 0; k < 16; ++k) {
    A = max16(A, min16(min16(lo4[k], lo4[(k + 4) & 15]), e[(k + 8) & 15]));
    B = min16(B, max16(max16(hi4[k], hi4[(k + 4) & 15]), e[(k + 8) & 15]));
  }
  const int best = std::max((int)A - 256, 256 - (int)B);
  return best > thr ? best - 1 : 0;
}
static uint32_t arcmax(const uint32_t (&g)[16]) {
  uint32_t suf0[8], pre1[8], suf1[8], pre0[8];
  suf0[7] = g[7]; for (int k = 6; k >= 0; --k) suf0[k] = min16(g[k], suf0[k + 1]);
  pre1[0] = g[8]; for (int j = 1; j < 8; ++j) pre1[j] = min16(g[8 + j], pre1[j - 1]);
  suf1[7] = g[15]; for (int k = 6; k >= 0; --k) suf1[k] = min16(g[8 + k], suf1[k + 1]);
  pre0[0] = g[0]; for (int j = 1; j < 8; ++j) pre0[j] = min16(g[j], pre0[j - 1]);
  uint32_t A = 0;
  for (int k = 0; k < 8; ++k) { A = max16(A, min16(suf0[k], pre1[k])); A = max16(A, min16(suf1[k], pre0[k])); }
  return A;
}
static int new_score(const uint32_t (&v)[16], int c, int thr, bool* amb) {
  const uint32_t hiA = max16(v[0], v[4]), loA = min16(v[0], v[4]), hiB = max16(v[8], v[12]), loB = min16(v[8], v[12]);
  const uint32_t X = min16(hiA, hiB), Y = max16(loA, loB);
  const int sec_large = (int)max16(X, Y), sec_small = (int)min16(X, Y);
  const bool cb = sec_large > c + thr, cd = sec_small < c - thr;
  const uint32_t mm = (cd && !cb) ? 0xFFu : 0u;
  uint32_t g[16];
  for (int k = 0; k < 16; ++k) g[k] = v[k] ^ mm;
  int best = (int)arcmax(g) - (int)((uint32_t)c ^ mm);
  *amb = cb && cd;
  if (cb && cd) {
    for (int k = 0; k < 16; ++k) g[k] = v[k] ^ 0xFFu;
    best = std::max(best, (int)arcmax(g) - (int)((uint32_t)c ^ 0xFFu));
  }
  return ((cb || cd) && best > thr) ? best - 1 : 0;
}
int main(int argc, char** argv) {
  const long per_mode = argc > 1 ? atol(argv[1]) : 4000000;
  srand(11);
  long n = 0, bad = 0, namb = 0, ncorner = 0;
  for (int mode = 0; mode < 6; ++mode)
    for (long it = 0; it < per_mode; ++it) {
      const int thr = mode == 5 ? rand() % 255 : (mode & 1 ? 10 : 20);
      int c = rand() & 255;
      if (mode == 3) c = (rand() & 1) ? rand() % 12 : 255 - rand() % 12;  // saturated centres
      uint32_t e[16], v[16];
      int start = rand() & 15, len = rand() % 17, amp = (rand() % 120), sign = rand() & 1 ? 1 : -1;
      int start2 = rand() & 15, len2 = mode >= 2 ? rand() % 10 : 0;
      for (int k = 0; k < 16; ++k) {
        int x = c + (rand() % 21 - 10) * (mode == 4 ? 3 : 1);
        if (((k - start) & 15) < len) x = c + sign * (amp + rand() % 8);
        if (((k - start2) & 15) < len2) x = c - sign * (amp + rand() % 30);
        x = std::min(255, std::max(0, x));
        v[k] = x;
        e[k] = 256 + x - c;
      }
      bool amb;
      const int a = old_score(e, thr), b = new_score(v, c, thr, &amb);
      ++n; namb += amb; ncorner += a > 0;
      if (a != b) { if (++bad < 10) printf("diff thr %d c %d old %d new %d\n", thr, c, a, b); }
    }
  printf("cases %ld corners %ld ambiguous %ld mismatches %ld\n", n, ncorner, namb, bad);
  return bad != 0;
}
