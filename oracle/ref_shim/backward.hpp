// Stand-in for backward.hpp (stack traces; common/basics/basics.h:25): empty.  TEST INFRASTRUCTURE for oracle/_ref.
#pragma once
