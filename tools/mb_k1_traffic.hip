// mb_k1_traffic.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on K1's OWN access shapes (MI355X_MICROARCH.md,
// HBM section: "other access widths and WRITE_SIZE are uncalibrated: calibrate on a known byte count in your own access
// pattern"; round-3 verdict item 5).  Three probe kernels, one wave per "instance" like K1, a million instances each (far
// beyond L2 and the Infinity Cache), nothing reused:
//   probe_records  what load_records + the warm-start read do: lanes 0-26 read 27 doubles of a 256-byte request record,
//                  lanes 32-44 13 doubles of a 128-byte state record, lanes 0-8 a 72-byte warm-start row (8 B per lane)
//   probe_tile     what load_tile does: 27 rows x 32 bytes of a byte map as aligned dwords, at a scattered position
//   probe_write    what K2 does: a 48-byte command from lane 0, six state doubles (0-2, 10-12), a 72-byte warm-start row,
//                  a 24-byte velocity row
// Prints one JSON line with the bytes each kernel moves, counted three ways (useful bytes, 64-byte sectors touched, 128-byte
// lines touched).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) by
// tools/profile_k1_traffic.sh, which divides.     build: hipcc --offload-arch=gfx950 -O3 tools/mb_k1_traffic.hip -o tools/_build/mb_k1_traffic
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void probe_records(const double* problems, const double* states, const double* warm, double* sink) {
  const int lane = threadIdx.x;
  const size_t b = blockIdx.x;
  double v = 0.0;
  if (lane < 27) v += problems[b * 32 + lane];
  else if (lane >= 32 && lane < 45) v += states[b * 16 + lane - 32];
  if (lane < 9) v += warm[b * 9 + lane];
  if (v == 123456.789) sink[0] = v;   // (never: the loads must not be optimised away)
}

__global__ __launch_bounds__(64) void probe_tile(const uint8_t* map, long pitch, long rows, double* sink) {
  const int lane = threadIdx.x;
  const uint64_t h = (blockIdx.x + 1) * 0x9E3779B97F4A7C15ull;
  const long x0 = (long)((h >> 20) % (uint64_t)(pitch - 64)) & ~3l, y0 = (long)((h >> 40) % (uint64_t)(rows - 32));
  uint32_t seen = 0u;
  for (int idx = lane; idx < 27 * 8; idx += 64) {
    const int row = idx >> 3, col = idx & 7;
    seen |= *reinterpret_cast<const uint32_t*>(map + (y0 + row) * pitch + x0 + 4 * col);
  }
  if (seen == 0x12345678u) sink[0] = 1.0;
}

struct Cmd { double vel[3]; double cost; int32_t status, iterations, evaluations, flags; };
__global__ __launch_bounds__(64) void probe_write(Cmd* commands, double* states, double* warm, double* vel) {
  const int lane = threadIdx.x;
  const size_t b = blockIdx.x;
  if (lane == 0) { Cmd c; c.vel[0] = 1; c.vel[1] = 2; c.vel[2] = 3; c.cost = 4; c.status = 0; c.iterations = 5; c.evaluations = 6; c.flags = 0; commands[b] = c; }
  if (lane < 3 || (lane >= 10 && lane < 13)) states[b * 16 + lane] = (double)lane;
  if (lane < 9) warm[b * 9 + lane] = (double)lane;
  if (lane < 3) vel[b * 3 + lane] = (double)lane;
}

// bytes of [off, off + len) counted in granules of g bytes
static double granules(size_t off, size_t len, size_t g) { return (double)(((off + len - 1) / g - off / g + 1) * g); }

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1u << 20);
  const long pitch = 16384, rows = 32768;   // 512 MB byte map
  double *problems, *states, *warm, *vel, *sink;
  uint8_t* map;
  Cmd* commands;
  CHECK(hipMalloc(&problems, n * 256)); CHECK(hipMalloc(&states, n * 128)); CHECK(hipMalloc(&warm, n * 72));
  CHECK(hipMalloc(&vel, n * 24)); CHECK(hipMalloc(&commands, n * 48)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&map, (size_t)pitch * rows));
  CHECK(hipMemset(problems, 0, n * 256)); CHECK(hipMemset(states, 0, n * 128)); CHECK(hipMemset(warm, 0, n * 72));
  CHECK(hipMemset(map, 0, (size_t)pitch * rows));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(probe_records, dim3(n), dim3(64), 0, 0, problems, states, warm, sink);
    hipLaunchKernelGGL(probe_tile, dim3(n), dim3(64), 0, 0, map, pitch, rows, sink);
    hipLaunchKernelGGL(probe_write, dim3(n), dim3(64), 0, 0, commands, states, warm, vel);
    CHECK(hipDeviceSynchronize());
  }
  // what each kernel moves per launch
  double rec_useful = n * (216.0 + 104.0 + 72.0), rec64 = 0, rec128 = 0, wr_useful = n * (48.0 + 48.0 + 72.0 + 24.0), wr64 = 0, wr128 = 0;
  for (size_t b = 0; b < n; ++b) {
    rec64 += granules(b * 256, 216, 64) + granules(b * 128, 104, 64) + granules(b * 72, 72, 64);
    rec128 += granules(b * 256, 216, 128) + granules(b * 128, 104, 128) + granules(b * 72, 72, 128);
    wr64 += granules(b * 48, 48, 64) + granules(b * 128, 24, 64) + granules(b * 128 + 80, 24, 64) + granules(b * 72, 72, 64) + granules(b * 24, 24, 64);
    wr128 += granules(b * 48, 48, 128) + granules(b * 128, 24, 128) + granules(b * 128 + 80, 24, 128) + granules(b * 72, 72, 128) + granules(b * 24, 24, 128);
  }
  // (neighbouring instances share the granules of the packed arrays -- warm start, commands, velocities: counted once per
  // instance above, i.e. an upper bound when the lines stay in L2 between neighbouring waves; the arrays' sizes are the lower bound)
  const double rec_arrays = n * (256.0 + 128.0 + 72.0), wr_arrays = n * (48.0 + 128.0 + 72.0 + 24.0);
  printf("{\"instances\": %zu, \"probe_records\": {\"useful\": %.0f, \"sectors64\": %.0f, \"lines128\": %.0f, \"arrays\": %.0f}, "
         "\"probe_tile\": {\"useful\": %.0f, \"sectors64\": %.0f, \"note\": \"27 rows x 32 B at a 4-byte-aligned x: one or two 64-byte sectors per row\"}, "
         "\"probe_write\": {\"useful\": %.0f, \"sectors64\": %.0f, \"lines128\": %.0f, \"arrays\": %.0f}}\n",
         n, rec_useful, rec64, rec128, rec_arrays, n * 27.0 * 32.0, n * 27.0 * 64.0 * 1.4375, wr_useful, wr64, wr128, wr_arrays);
  return 0;
}
