// forwarding stand-in: the adapters include the reference header of this name (tests/test_adapter_link.py)
#include "esl_ref_surface.hpp"
