// Readers of the reference's shipped data formats (ROS/PCL-free input side of the real-world and consistency drivers,
// "next" row N2 of SURVEY.md 8f): src/benchmark/benchmark_realworld.cpp:31-106 read_pose / read_file.  Host-only.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" {

// alidarPose.csv (benchmark_realworld.cpp:31-73): 4 text lines per pose, rows of [R|t], element (3,3) is the
// timestamp.  poses: up to max_poses * 12 doubles (R column-major, p); stamps optional.  Returns #poses, <0 on error.
int balm_read_pose_csv(const char *path, int max_poses, double *poses, double *stamps) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  std::vector<double> nums;
  double v;
  int ch;
  while (fscanf(f, "%lf", &v) == 1) {
    nums.push_back(v);
    do { ch = fgetc(f); } while (ch == ',' || ch == ' ' || ch == '\r' || ch == '\n');
    if (ch != EOF) ungetc(ch, f);
  }
  fclose(f);
  int W = (int)(nums.size() / 16);
  if (W > max_poses) W = max_poses;
  for (int m = 0; m < W; m++) {
    double *q = poses + 12 * m;
    for (int r = 0; r < 3; r++) {
      for (int c = 0; c < 3; c++) q[3 * c + r] = nums[16 * m + 4 * r + c];
      q[9 + r] = nums[16 * m + 4 * r + 3];
    }
    if (stamps) stamps[m] = nums[16 * m + 15];
  }
  return W;
}

// binary PCD with FIELDS x y z ... (all 4-byte floats; the shipped files have 8 fields = 32-byte records).
// Pass xyz = NULL to query the point count.  Returns #points written, <0 on error.
long balm_read_pcd_xyz(const char *path, float *xyz, long max_points) {
  FILE *f = fopen(path, "rb");
  if (!f) return -1;
  char line[512];
  long npts = -1;
  int nfields = 0;
  bool binary = false;
  while (fgets(line, sizeof line, f)) {
    if (!strncmp(line, "FIELDS", 6)) { for (char *p = line + 6; *p; p++) if (*p == ' ' && p[1] != ' ' && p[1] != '\n') nfields++; }
    if (!strncmp(line, "POINTS", 6)) npts = atol(line + 7);
    if (!strncmp(line, "DATA", 4)) { binary = !strncmp(line + 5, "binary", 6) && strncmp(line + 5, "binary_compressed", 17); break; }
  }
  if (npts < 0 || nfields < 3 || !binary) { fclose(f); return -2; }
  if (!xyz) { fclose(f); return npts; }
  if (npts > max_points) npts = max_points;
  std::vector<float> rec((size_t)nfields * 4096);
  long done = 0;
  while (done < npts) {
    const long want = std::min<long>(4096, npts - done);
    const size_t got = fread(rec.data(), sizeof(float) * nfields, (size_t)want, f);
    for (size_t k = 0; k < got; k++) {
      xyz[3 * (done + k)] = rec[k * nfields]; xyz[3 * (done + k) + 1] = rec[k * nfields + 1]; xyz[3 * (done + k) + 2] = rec[k * nfields + 2];
    }
    done += (long)got;
    if ((long)got < want) break;
  }
  fclose(f);
  return done;
}

}  // extern "C"
