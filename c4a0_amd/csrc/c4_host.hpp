// c4_host.hpp -- host-side helpers shared by the translation units of libc4a0_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <mutex>
#include <string>
#include <utility>

namespace c4host {

// The one error string behind c4_last_error_string() (thread-local, defined in c4_session.hip).
int fail(int code, const std::string& msg);

// Entry points run on the device of their session / stream and leave the caller's current
// device as they found it (PyTorch keeps its own notion of the current device).
class DeviceGuard {
 public:
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&prev_) != hipSuccess) prev_ = -1;
    if (device >= 0 && device != prev_) { err_ = hipSetDevice(device); switched_ = err_ == hipSuccess; }
  }
  ~DeviceGuard() {
    if (switched_ && prev_ >= 0) (void)hipSetDevice(prev_);
  }
  hipError_t error() const { return err_; }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;

 private:
  int prev_ = -1;
  bool switched_ = false;
  hipError_t err_ = hipSuccess;
};

// Device a stream belongs to (-1 = the null stream: the caller's current device).
inline int stream_device(hipStream_t stream) {
  int dev = -1;
  if (stream != nullptr && hipStreamGetDevice(stream, &dev) != hipSuccess) dev = -1;
  if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;
  return dev;
}

// More than 64 KB of dynamic LDS needs an opt-in per kernel AND per device (hipFuncSetAttribute
// acts on the current device's function object): remembered per (kernel, device).
inline hipError_t opt_in_lds(const void* kernel, int bytes, int device) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, bool> done;
  std::lock_guard<std::mutex> lock(mu);
  const auto key = std::make_pair(kernel, device);
  if (done.count(key)) return hipSuccess;
  const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done[key] = true;
  return e;
}

}  // namespace c4host
