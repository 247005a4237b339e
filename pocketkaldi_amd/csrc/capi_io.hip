// capi_io.hip -- files and single-utterance front-end entries: the NNT0 / LAY0 / MAT0 / VEC0 model files
// (nnet.cc:80-147), pk_load's key = value file (pocketkaldi.cc:72-144), strict 16 kHz WAV ingestion
// (pcm_reader.cc:45-220), Fbank::Compute / CMVN for one utterance, and the parity-test hooks.
#include <hip/hip_runtime.h>
#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <cmath>

#include <algorithm>
#include <string>
#include <mutex>
#include <utility>
#include <unordered_set>
#include <vector>

#include "pk_host.h"

using namespace pkmi;
using namespace pkhost;

extern "C" {

int pk_mi355_am_read(pk_mi355_am_t *am, const char *nnet_path, const char *prior_path,
                     const char *tid2pdf_path, int left_context, int right_context,
                     int num_pdfs) {
  if (!am || am->finalized) return Fail(PK_MI355_E_STATE, "model already finalized");
  FileBuf f;
  int rc = f.Open(nnet_path);
  if (rc) return rc;
  int32_t sec, num_layers;
  if (!f.Tag("NNT0") || !f.I32(&sec) || !f.I32(&num_layers) || sec != 4)
    return Fail(PK_MI355_E_IO, "NNT0 section expected in %s", nnet_path);
  for (int l = 0; l < num_layers; ++l) {
    int32_t type;
    if (!f.Tag("LAY0") || !f.I32(&sec) || !f.I32(&type))
      return Fail(PK_MI355_E_IO, "LAY0 section expected in %s", nnet_path);
    if (sec != 4)    // nnet.cc:94-101
      return Fail(PK_MI355_E_IO, "read_layer: section_size == 4 expected, but %d found (%s)", sec, nnet_path);
    if (type == PK_NNET_LINEAR_LAYER) {
      int32_t rows, cols;
      if (!f.Tag("MAT0") || !f.I32(&sec) || !f.I32(&rows) || !f.I32(&cols) || rows <= 0 || cols <= 0)
        return Fail(PK_MI355_E_IO, "MAT0 section expected in %s", nnet_path);
      std::vector<float> W((size_t)rows * cols), row, bias;
      for (int r = 0; r < rows; ++r) {
        if ((rc = f.Vec(&row))) return rc;
        if ((int)row.size() != cols)
          return Fail(PK_MI355_E_IO, "Matrix::Read: row dim %d expected, but %d found: %s", cols, (int)row.size(), nnet_path);
        memcpy(&W[(size_t)r * cols], row.data(), sizeof(float) * cols);
      }
      if ((rc = f.Vec(&bias))) return rc;
      if ((int)bias.size() != rows) return Fail(PK_MI355_E_IO, "bias dimension mismatch in %s", nnet_path);
      if ((rc = pk_mi355_am_add_linear(am, cols, rows, W.data(), bias.data()))) return rc;
    } else {
      if ((rc = pk_mi355_am_add_layer(am, type)))
        return Fail(PK_MI355_E_IO, "read_layer: unexpected layer type: %d (%s)", type, nnet_path);
    }
  }
  std::vector<float> prior;
  FileBuf pf;
  if ((rc = pf.Open(prior_path)) || (rc = pf.Vec(&prior))) return rc;
  std::vector<int32_t> tid;
  if (tid2pdf_path) {
    FileBuf tf;
    if ((rc = tf.Open(tid2pdf_path)) || (rc = tf.Vec(&tid))) return rc;
  }
  if ((int)prior.size() != num_pdfs)
    return Fail(PK_MI355_E_INVALID, "prior has %d entries, num_pdfs = %d", (int)prior.size(), num_pdfs);
  return pk_mi355_am_finalize(am, prior.data(), num_pdfs, left_context, right_context,
                              tid.empty() ? nullptr : tid.data(), (int)tid.size());
}

}  // extern "C"

// ---- pk_load's share of this path (pocketkaldi.cc:72-144): the key = value model file.
namespace {

std::string TrimWs(const std::string &s) {
  size_t a = 0, b = s.size();
  while (a < b && isspace((unsigned char)s[a])) ++a;
  while (b > a && isspace((unsigned char)s[b - 1])) --b;
  return s.substr(a, b - a);
}

// configuration.cc:16-55: '#' comments and blank lines skipped, exactly one '=' per line,
// keys lower-cased, empty values rejected.
struct ConfigFile {
  std::string filename;
  std::vector<std::pair<std::string, std::string>> table;

  int Read(const char *path) {
    filename = path;
    FILE *f = fopen(path, "r");
    if (!f) return Fail(PK_MI355_E_IO, "cannot open %s", path);
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) text.append(buf, n);
    fclose(f);
    size_t pos = 0;
    while (pos < text.size()) {
      size_t e = text.find('\n', pos);
      if (e == std::string::npos) e = text.size();
      const std::string line = TrimWs(text.substr(pos, e - pos));
      pos = e + 1;
      if (line.empty() || line[0] == '#') continue;
      const size_t eq = line.find('=');
      if (eq == std::string::npos || line.find('=', eq + 1) != std::string::npos)
        return Fail(PK_MI355_E_IO, "Unexpected line in %s: %s", path, line.c_str());
      std::string key = TrimWs(line.substr(0, eq));
      const std::string value = TrimWs(line.substr(eq + 1));
      for (auto &c : key) c = (char)tolower((unsigned char)c);
      if (value.empty()) return Fail(PK_MI355_E_IO, "Value cound not be empty: %s: %s", path, line.c_str());
      bool found = false;
      for (auto &kv : table)
        if (kv.first == key) { kv.second = value; found = true; }
      if (!found) table.emplace_back(key, value);
    }
    return 0;
  }
  const std::string *Find(const char *key) const {
    for (const auto &kv : table)
      if (kv.first == key) return &kv.second;
    return nullptr;
  }
  // configuration.cc:57-72: relative paths are relative to the directory of the file
  int Path(const char *key, std::string *out) const {
    const std::string *v = Find(key);
    if (!v) return Fail(PK_MI355_E_IO, "Unable to find key '%s' in %s", key, filename.c_str());
    const size_t slash = filename.rfind('/');
    *out = ((*v)[0] == '/' || slash == std::string::npos) ? *v : filename.substr(0, slash + 1) + *v;
    return 0;
  }
  int Integer(const char *key, int *out) const {
    const std::string *v = Find(key);
    if (!v) return Fail(PK_MI355_E_IO, "Unable to find key '%s' in %s", key, filename.c_str());
    char *end = nullptr;
    const long x = strtol(v->c_str(), &end, 10);
    if (end == v->c_str()) return Fail(PK_MI355_E_IO, "key '%s' in %s is not an integer: %s", key, filename.c_str(), v->c_str());
    *out = (int)x;
    return 0;
  }
};

}  // namespace

extern "C" {

int pk_mi355_load(const char *config_path, int precision, pk_mi355_am_t **am_out, float *cmvn_stats41) {
  if (!config_path || !am_out || !cmvn_stats41) return Fail(PK_MI355_E_INVALID, "null argument");
  *am_out = nullptr;
  ConfigFile conf;
  int rc = conf.Read(config_path);
  if (rc) return rc;
  // CMVN global statistics, pocketkaldi.cc:101-116: VEC0 of 40 sums + the frame count
  std::string cmvn_path, nnet, prior, tid2pdf;
  if ((rc = conf.Path("cmvn_stats", &cmvn_path))) return rc;
  FileBuf cf;
  std::vector<float> stats;
  if ((rc = cf.Open(cmvn_path.c_str())) || (rc = cf.Vec(&stats))) return rc;
  if ((int)stats.size() != kNumBins + 1)
    return Fail(PK_MI355_E_IO, "cmvn_stats in %s has %d entries, %d expected", cmvn_path.c_str(), (int)stats.size(), kNumBins + 1);
  // AcousticModel::Read, am.cc:22-62 (a missing left_context is not an error there either)
  int left = 0, right = 0, num_pdfs = 0;
  if ((rc = conf.Path("nnet", &nnet)) || (rc = conf.Path("prior", &prior))) return rc;
  if (conf.Find("left_context") && (rc = conf.Integer("left_context", &left))) return rc;
  if ((rc = conf.Integer("right_context", &right)) || (rc = conf.Integer("num_pdfs", &num_pdfs)) ||
      (rc = conf.Path("tid2pdf", &tid2pdf)))
    return rc;
  if (left < 0 || right < 0) return Fail(PK_MI355_E_INVALID, "negative context in %s", config_path);
  pk_mi355_am_t *am = pk_mi355_am_create();
  if (!am) return PK_MI355_E_DEVICE;
  if ((rc = pk_mi355_am_set_precision(am, precision)) ||
      (rc = pk_mi355_am_read(am, nnet.c_str(), prior.c_str(), tid2pdf.c_str(), left, right, num_pdfs))) {
    pk_mi355_am_destroy(am);
    return rc;
  }
  memcpy(cmvn_stats41, stats.data(), sizeof(float) * (kNumBins + 1));
  *am_out = am;
  return 0;
}

int pk_mi355_test_logf(const float *x, int n, float *out) {
  if (!x || !out || n < 0) return Fail(PK_MI355_E_INVALID, "bad argument");
  int rc = UseDevice(CurrentDevice());
  if (rc || n == 0) return rc;
  FrontendTables host;
  if (BuildFrontendTables(&host)) return Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  FrontendTables *d_tab = nullptr;
  float *d_x = nullptr, *d_y = nullptr;
  hipError_t e = hipMalloc(&d_tab, sizeof(host));
  if (e == hipSuccess) e = hipMalloc(&d_x, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc(&d_y, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMemcpy(d_tab, &host, sizeof(host), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_x, x, sizeof(float) * (size_t)n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    LaunchLogfTest(d_x, n, d_tab, d_y, nullptr);
    e = hipMemcpy(out, d_y, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost);
  }
  hipFree(d_tab); hipFree(d_x); hipFree(d_y);
  if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "test_logf: %s", hipGetErrorString(e));
  return 0;
}

int pk_mi355_test_srfft512(const float *frames, int num_frames, float *spectra) {
  if (!frames || !spectra || num_frames < 0) return Fail(PK_MI355_E_INVALID, "bad argument");
  int rc = UseDevice(CurrentDevice());
  if (rc || num_frames == 0) return rc;
  FrontendTables host;
  if (BuildFrontendTables(&host)) return Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  const size_t bytes = sizeof(float) * (size_t)num_frames * kFftSize;
  FrontendTables *d_tab = nullptr;
  float *d_x = nullptr, *d_y = nullptr;
  hipError_t e = hipMalloc(&d_tab, sizeof(host));
  if (e == hipSuccess) e = hipMalloc(&d_x, bytes);
  if (e == hipSuccess) e = hipMalloc(&d_y, bytes);
  if (e == hipSuccess) e = hipMemcpy(d_tab, &host, sizeof(host), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_x, frames, bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    LaunchSrfft512Test(d_x, num_frames, d_tab, d_y, nullptr);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(spectra, d_y, bytes, hipMemcpyDeviceToHost);
  hipFree(d_tab); hipFree(d_x); hipFree(d_y);
  if (e != hipSuccess) return Fail(PK_MI355_E_DEVICE, "test_srfft512: %s", hipGetErrorString(e));
  return 0;
}

// ------------------------------------------------------------------ single-utterance front-end

int pk_mi355_fbank_compute(const pk_vector_t *wave, pk_matrix_t *out) {
  if (!wave || !out) return Fail(PK_MI355_E_INVALID, "null argument");
  int rc = UseDevice(CurrentDevice());
  if (rc) return rc;
  const int T = pk_mi355_num_frames(wave->dim);
  if (T == 0) return ResizeHostMatrix(out, 0, 0);          // fbank.cc:272-273
  if ((rc = ResizeHostMatrix(out, kNumBins, T))) return rc;
  FrontendTables host;
  if (BuildFrontendTables(&host)) return Fail(PK_MI355_E_INVALID, "front-end table construction failed");
  FrontendTables *d_tab = nullptr;
  float *d_wave = nullptr, *d_raw = nullptr;
  int64_t *d_i64 = nullptr;
  int32_t *d_T = nullptr;
  int64_t zeros[2] = {0, 0};
  int32_t hT = T;
  int ret = 0;
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
  step(hipMalloc(&d_tab, sizeof(FrontendTables)));
  step(hipMalloc(&d_wave, sizeof(float) * wave->dim));
  step(hipMalloc(&d_raw, sizeof(float) * (size_t)T * kNumBins));
  step(hipMalloc(&d_i64, sizeof(int64_t) * 2));
  step(hipMalloc(&d_T, sizeof(int32_t)));
  if (e == hipSuccess) {
    step(hipMemcpy(d_tab, &host, sizeof(FrontendTables), hipMemcpyHostToDevice));
    step(hipMemcpy(d_wave, wave->data, sizeof(float) * wave->dim, hipMemcpyHostToDevice));
    step(hipMemcpy(d_i64, zeros, sizeof(zeros), hipMemcpyHostToDevice));
    step(hipMemcpy(d_T, &hT, sizeof(hT), hipMemcpyHostToDevice));
  }
  if (e == hipSuccess) {
    UttLayout lay{d_i64, d_T, d_i64 + 1, d_i64 + 1};
    LaunchFbank(d_wave, nullptr, lay, 1, T, d_tab, d_raw, nullptr);
    step(hipGetLastError());
    step(hipMemcpy(out->data, d_raw, sizeof(float) * (size_t)T * kNumBins, hipMemcpyDeviceToHost));
  }
  if (e != hipSuccess) ret = Fail(PK_MI355_E_DEVICE, "fbank_compute: %s", hipGetErrorString(e));
  hipFree(d_tab); hipFree(d_wave); hipFree(d_raw); hipFree(d_i64); hipFree(d_T);
  return ret;
}

int pk_mi355_cmvn_apply(const pk_vector_t *global_stats, const pk_matrix_t *raw, pk_matrix_t *out) {
  if (!global_stats || !raw || !out) return Fail(PK_MI355_E_INVALID, "null argument");
  if (global_stats->dim != kNumBins + 1 || (raw->ncol > 0 && raw->nrow != kNumBins))
    return Fail(PK_MI355_E_INVALID, "cmvn expects 41 global stats and 40-dim features");
  int rc = UseDevice(CurrentDevice());
  if (rc) return rc;
  const int T = raw->ncol;
  if ((rc = ResizeHostMatrix(out, T > 0 ? kNumBins : 0, T))) return rc;
  if (T == 0) return 0;
  float *d_raw = nullptr, *d_g = nullptr, *d_yt = nullptr;
  CmvnTables *d_ctab = nullptr;
  CmvnTables ctab;
  BuildCmvnTables(global_stats->data[kNumBins], &ctab);
  int64_t *d_i64 = nullptr;
  int32_t *d_T = nullptr;
  int64_t zeros[2] = {0, 0};
  int32_t hT = T;
  const int64_t ld = RoundUp(T, 64);
  std::vector<float> tmp((size_t)kNumBins * T);
  hipError_t e = hipSuccess;
  auto step = [&](hipError_t x) { if (e == hipSuccess) e = x; };
  float *d_raw_alloc = nullptr;
  const size_t raw_floats = (size_t)T * kNumBins + kCmvnRawLead + kCmvnRawSlack;
  step(hipMalloc(&d_raw_alloc, sizeof(float) * raw_floats));
  if (e == hipSuccess) step(hipMemset(d_raw_alloc, 0, sizeof(float) * raw_floats));
  d_raw = d_raw_alloc ? d_raw_alloc + kCmvnRawLead : nullptr;
  step(hipMalloc(&d_g, sizeof(float) * (kNumBins + 1)));
  step(hipMalloc(&d_ctab, sizeof(CmvnTables)));
  step(hipMalloc(&d_yt, sizeof(float) * ld * kNumBins));
  step(hipMalloc(&d_i64, sizeof(int64_t) * 2));
  step(hipMalloc(&d_T, sizeof(int32_t)));
  if (e == hipSuccess) {
    step(hipMemcpy(d_raw, raw->data, sizeof(float) * (size_t)T * kNumBins, hipMemcpyHostToDevice));
    step(hipMemcpy(d_g, global_stats->data, sizeof(float) * (kNumBins + 1), hipMemcpyHostToDevice));
    step(hipMemcpy(d_ctab, &ctab, sizeof(CmvnTables), hipMemcpyHostToDevice));
    step(hipMemcpy(d_i64, zeros, sizeof(zeros), hipMemcpyHostToDevice));
    step(hipMemcpy(d_T, &hT, sizeof(hT), hipMemcpyHostToDevice));
  }
  if (e == hipSuccess) {
    UttLayout lay{d_i64, d_T, d_i64, d_i64 + 1};
    LaunchCmvn(d_raw, lay, 1, d_g, d_ctab, 0, 0, d_yt, ld, nullptr);
    step(hipGetLastError());
    step(hipMemcpy2D(tmp.data(), sizeof(float) * T, d_yt, sizeof(float) * ld, sizeof(float) * T, kNumBins,
                     hipMemcpyDeviceToHost));
  }
  int ret = 0;
  if (e != hipSuccess) ret = Fail(PK_MI355_E_DEVICE, "cmvn_apply: %s", hipGetErrorString(e));
  else
    for (int t = 0; t < T; ++t)
      for (int d = 0; d < kNumBins; ++d) out->data[(size_t)t * kNumBins + d] = tmp[(size_t)d * T + t];
  hipFree(d_raw_alloc); hipFree(d_g); hipFree(d_ctab); hipFree(d_yt); hipFree(d_i64); hipFree(d_T);
  return ret;
}

// ------------------------------------------------------------------ pk_process, acoustic half

// pcm_reader.cc:45-220: strict 44-byte-header RIFF/WAVE PCM, mono, 16 kHz, 8/16/32-bit,
// sample values kept unscaled as float.
int pk_mi355_16kpcm_read(const char *filename, pk_vector_t *pcm_data) {
  if (!filename || !pcm_data) return Fail(PK_MI355_E_INVALID, "null argument");
  FileBuf f;
  int rc = f.Open(filename);
  if (rc) return rc;
  const unsigned char *b = f.d.data();
  const long size = (long)f.d.size();
  auto i32 = [&](long off) { int32_t v; memcpy(&v, b + off, 4); return v; };
  auto i16 = [&](long off) { int16_t v; memcpy(&v, b + off, 2); return (int)v; };
  if (size < 44) return Fail(PK_MI355_E_IO, "file too short for a WAVE header: %s", filename);
  if (memcmp(b, "RIFF", 4)) return Fail(PK_MI355_E_IO, "chunk_name == 'RIFF' expected: %s", filename);
  if (i32(4) != size - 8) return Fail(PK_MI355_E_IO, "chunk_size == %ld expected, but %d found: %s", size - 8, i32(4), filename);
  if (memcmp(b + 8, "WAVE", 4)) return Fail(PK_MI355_E_IO, "Format == 'WAVE' expected: %s", filename);
  if (memcmp(b + 12, "fmt ", 4)) return Fail(PK_MI355_E_IO, "subchunk1 == 'fmt ' expected: %s", filename);
  if (i32(16) != 16) return Fail(PK_MI355_E_IO, "subchunk1_size == 16 expected, but %d found: %s", i32(16), filename);
  if (i16(20) != 1) return Fail(PK_MI355_E_IO, "audio_format == 1 (PCM) expected, but %d found: %s", i16(20), filename);
  if (i16(22) != 1) return Fail(PK_MI355_E_IO, "num_channels == 1 (mono) expected, but %d found: %s", i16(22), filename);
  const int rate = i32(24);
  if (rate != kSampleRate) return Fail(PK_MI355_E_IO, "sample_rate == 16000 expected, but %d found: %s", rate, filename);
  const int byte_rate = i32(28), align = i16(32), bits = i16(34);
  if (bits != 8 && bits != 16 && bits != 32)
    return Fail(PK_MI355_E_IO, "bits_per_sample == 8, 16 or 32 expected, but %d found: %s", bits, filename);
  if (byte_rate != rate * bits / 8) return Fail(PK_MI355_E_IO, "bytes_rate == %d expected, but %d found: %s", rate * bits / 8, byte_rate, filename);
  if (align != bits / 8) return Fail(PK_MI355_E_IO, "block_align == %d expected, but %d found: %s", bits / 8, align, filename);
  if (memcmp(b + 36, "data", 4)) return Fail(PK_MI355_E_IO, "subchunk2 == 'data' expected: %s", filename);
  if (i32(40) != size - 44) return Fail(PK_MI355_E_IO, "subchunk2_size == %ld expected, but %d found: %s", size - 44, i32(40), filename);
  const int n = (int)((size - 44) / (bits / 8));
  float *s = static_cast<float *>(realloc(pcm_data->data, sizeof(float) * (n > 0 ? n : 1)));
  if (!s) return Fail(PK_MI355_E_INVALID, "out of host memory");
  const unsigned char *p = b + 44;
  for (int i = 0; i < n; ++i) {
    if (bits == 8) { s[i] = (float)(int8_t)p[0]; p += 1; }
    else if (bits == 16) { int16_t v; memcpy(&v, p, 2); s[i] = (float)v; p += 2; }
    else { int32_t v; memcpy(&v, p, 4); s[i] = (float)v; p += 4; }
  }
  pcm_data->data = s;
  pcm_data->dim = n;
  return 0;
}

}  // extern "C"
