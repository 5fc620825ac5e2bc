// config.h -- numeric constants shared by the host-side BA library.
// Same names and values as reference batotp/config.h:25-32 so user code that includes it keeps
// compiling.
#ifndef BATOTP_AMD_CONFIG_H
#define BATOTP_AMD_CONFIG_H

#define _CONFIG_FILE "./input/config.dat"

static const double PI = 3.14159265358979323846;
static const double _DEG2RAD = PI / 180.0;
static const double _RAD2DEG = 180.0 / PI;
static const double _g = 9.81;

static const int _MAXCHAR = 100;

#endif
