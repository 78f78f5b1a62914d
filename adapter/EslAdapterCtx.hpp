// EslAdapterCtx.hpp — one lazily created esl_ctx shared by the three adapter classes, behind a mutex.
// A context is re-entrant per context but not thread-safe inside one (include/esl.h), and Tracking may be driven from
// a different thread than the viewer: every adapter call takes the lock for the duration of its C-ABI calls.
#pragma once
#include <iostream>
#include <mutex>

#include "esl.h"

namespace esl_adapter {

inline std::mutex& CtxMutex() {
  static std::mutex m;
  return m;
}

// call with CtxMutex() held; returns nullptr (after reporting to std::cerr, as the reference reports errors) when no
// HIP device is usable -- there is no CPU fallback behind the adapters
inline esl_ctx* SharedCtx() {
  static esl_ctx* ctx = nullptr;
  static bool tried = false;
  if (!ctx && !tried) {
    tried = true;
    int device = 0;
    if (esl_ctx_create(device, &ctx) != ESL_OK) {
      std::cerr << "esl: " << esl_last_error() << std::endl;
      ctx = nullptr;
    }
  }
  return ctx;
}

}  // namespace esl_adapter
