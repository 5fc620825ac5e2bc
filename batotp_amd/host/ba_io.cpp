// ba_io.cpp -- configuration and trajectory file IO of BA (host only, outside the GPU hot path).
//
// Serves the file formats of the reference behind BA's public IO methods
// (reference batotp/ba.cpp:1942-2087 readConfigData, 2100-2197 loadConfigData, 2206-2245 loadTrajectoryData,
// 2257-2461 trajReadBIN / trajReadCSV, 2510-2528 writeOutputData, 2582-2759 trajWriteBIN / trajWriteCSV / sdotWrite).
// The formats and the messages are the reference's; the readers are this repository's own design:
//   * config.dat is read into memory once and walked with a line cursor; WHAT is read is a table of field
//     descriptors bound to BA's members (one row per line of the file, section breaks as rows), so the parser
//     itself is a dozen lines and the item count the reference checks (34 + 4 nJoints) falls out of the table;
//   * CSV trajectories are split into rows and cells in memory, the header row decides which column groups exist;
//   * binary files go through two small typed helpers (PodReader / F32RowWriter).
#include <clocale>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

#include "ba.h"
#include "util.h"

namespace BATOTP
{

namespace
{
// the text formats use '.' decimals: pin LC_NUMERIC while a text file is parsed or written (the reference switches
// the process locale for good, ba.cpp:1944-1945; here it is restored)
class NumericLocale
{
public:
   NumericLocale()
   {
      const char *cur = std::setlocale(LC_NUMERIC, NULL);
      _saved = cur ? cur : "C";
      std::setlocale(LC_NUMERIC, "en_US.UTF-8");
   }
   ~NumericLocale() { std::setlocale(LC_NUMERIC, _saved.c_str()); }

private:
   std::string _saved;
};

int pathTypeFromString(const std::string &s)
{
   if (s == "JOINT") return JOINT;
   if (s == "CART") return CART;
   if (s == "BOTH") return BOTH;
   return 0;
}

// ---- config.dat ------------------------------------------------------------------------------------
// One physical line of the file = one item (a word, a number, or nJoints / 3 numbers); the rest of the line is a
// comment.  The file is held as lines; a cursor hands them out in order.
class LineCursor
{
public:
   explicit LineCursor(std::istream &in)
   {
      std::string l;
      while (std::getline(in, l)) _lines.push_back(l);
   }
   void skip(int n) { _at += (size_t)n; }
   // whitespace-separated words of the next line (empty when the file has ended)
   std::vector<std::string> words()
   {
      std::vector<std::string> w;
      if (_at < _lines.size())
      {
         std::istringstream ss(_lines[_at]);
         std::string t;
         while (ss >> t) w.push_back(t);
      }
      ++_at;
      return w;
   }

private:
   std::vector<std::string> _lines;
   size_t _at = 0;
};

// what a line of config.dat holds and where it goes
struct Field
{
   enum Kind { SKIP, WORD, FLAG, INT, UINT, REAL, REALS } kind;
   void *dst;       // std::string / bool / int / unsigned / double / std::vector<double>
   int count;       // SKIP: lines to skip; REALS: numbers on the line (-1: one per joint)
};
Field skipLines(int n) { return Field{Field::SKIP, nullptr, n}; }
Field word(std::string &d) { return Field{Field::WORD, &d, 1}; }
Field flag(bool &d) { return Field{Field::FLAG, &d, 1}; }
Field integer(int &d) { return Field{Field::INT, &d, 1}; }
Field count(unsigned int &d) { return Field{Field::UINT, &d, 1}; }
Field real(double &d) { return Field{Field::REAL, &d, 1}; }
Field reals(std::vector<double> &d, int n) { return Field{Field::REALS, &d, n}; }

bool toDouble(const std::string &s, double &v)
{
   char *end = nullptr;
   v = std::strtod(s.c_str(), &end);   // also "NAN", as the reference's %lf does
   return end != s.c_str();
}

// reads one field; returns the number of items converted (what fscanf would have counted)
int parseField(const Field &f, LineCursor &cur, unsigned int nJoints)
{
   if (f.kind == Field::SKIP) { cur.skip(f.count); return 0; }
   const std::vector<std::string> w = cur.words();
   const int want = f.kind == Field::REALS ? (f.count < 0 ? (int)nJoints : f.count) : 1;
   int got = 0;
   double v = 0;
   switch (f.kind)
   {
   case Field::WORD:
      if (!w.empty()) { *static_cast<std::string *>(f.dst) = w[0]; got = 1; }
      break;
   case Field::FLAG:
      if (!w.empty() && toDouble(w[0], v)) { *static_cast<bool *>(f.dst) = ((int)v == 1); got = 1; }
      break;
   case Field::INT:
      if (!w.empty() && toDouble(w[0], v)) { *static_cast<int *>(f.dst) = (int)v; got = 1; }
      break;
   case Field::UINT:
      if (!w.empty() && toDouble(w[0], v)) { *static_cast<unsigned int *>(f.dst) = (unsigned int)v; got = 1; }
      break;
   case Field::REAL:
      if (!w.empty() && toDouble(w[0], v)) { *static_cast<double *>(f.dst) = v; got = 1; }
      break;
   case Field::REALS:
   {
      std::vector<double> &out = *static_cast<std::vector<double> *>(f.dst);
      out.assign((size_t)want, 0.0);
      for (int k = 0; k < want && k < (int)w.size(); ++k)
      {
         if (!toDouble(w[k], out[k])) break;
         ++got;
      }
      break;
   }
   default: break;
   }
   return got;
}

// ---- binary helpers ----------------------------------------------------------------------------------
class PodReader
{
public:
   explicit PodReader(FILE *f) : _f(f) {}
   template <typename T>
   void get(T &v) { _items += fread(&v, sizeof(T), 1, _f); }
   // n float32 values widened into a row of doubles
   void row(std::vector<double> &dst, size_t n)
   {
      _tmp.resize(n);
      _items += fread(_tmp.data(), sizeof(float), n, _f);
      dst.assign(_tmp.begin(), _tmp.end());
   }
   size_t items() const { return _items; }

private:
   FILE *_f;
   size_t _items = 0;
   std::vector<float> _tmp;
};

class F32RowWriter
{
public:
   explicit F32RowWriter(FILE *f) : _f(f) {}
   template <typename T>
   void put(const T &v) { fwrite(&v, sizeof(T), 1, _f); }
   void rows(const std::vector<std::vector<double>> &ch, size_t nRows)
   {
      for (size_t j = 0; j < nRows; ++j) row(ch[j]);
   }
   void row(const std::vector<double> &v)
   {
      _tmp.assign(v.begin(), v.end());      // double -> float32, the file's precision
      fwrite(_tmp.data(), sizeof(float), _tmp.size(), _f);
   }

private:
   FILE *_f;
   std::vector<float> _tmp;
};
} // namespace

// ---------------------------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------------------------
// what both entry points do once the members are filled (reference ba.cpp:2020-2028, 2048, 2063-2082)
int BA::finishConfig()
{
   _isParallelMechOrig = _isParallelMech;
   _robotType = myRobot.call_set_robotType(_robotTypeStr);
   _isGenericRobot = (_robotTypeStr == "GENJNT");
   if (_robotType == 0)
   {
      printf("\nreadInputData() error: robotType is %s", _robotTypeStr.c_str());
      printf("It should be 'KUKA', 'UR', 'RR', 'CSPR3DOF', or 'GENJNT'.\n");
      return -1;
   }
   if (_pathType == 0)
   {
      printf("\nreadInputData() error: pathType is %s", _pathTypeStr.c_str());
      printf("It should be 'JOINT', 'CART', or 'BOTH'.\n");
      return -1;
   }
   for (unsigned int j = 0; j < _nJoints && j < _JntTrqMin.size() && j < _JntTrqMax.size(); ++j)
      if (std::isnan(_JntTrqMin[j])) _JntTrqMin[j] = -_JntTrqMax[j];   // NAN lower torque limit = symmetric limits
   _quadraticRadThresh = _cartThresh * _cartThresh;
   if (_sWeights.size() < 3) _sWeights.resize(3, 0.0);
   const double wSum = _sWeights[0] + _sWeights[1] + _sWeights[2];
   if (wSum <= 0)
   {
      printf("Error in readInputData(): sum(sWeights) should be greater than 0.\n");
      return -1;
   }
   for (int k = 0; k < 3; ++k) _sWeights[k] /= wSum;
   return 0;
}

int BA::readConfigData(const char *filename)
{
   NumericLocale pin;
   std::ifstream in(filename);
   if (!in)
   {
      printf("\nUnable to open file %s\n", filename);
      return -1;
   }
   printf("\nConfiguration file: '%s'\n", filename);
   LineCursor cur(in);

   // the file, line by line (input/README_for_config_file.txt; example input/RR/config.dat:1-45)
   std::string trajName;
   const int perJoint = -1;
   const Field layout[] = {
      skipLines(3),
      word(_robotTypeStr), flag(_isParallelMech), count(_nJoints), count(_nCart), word(trajName), flag(_isBINfile), word(_pathTypeStr),
      skipLines(2),
      flag(_areJointAnglesDegrees), flag(_isJntVelConOn), reals(_JntVelMax, perJoint), flag(_isJntAccConOn), reals(_JntAccMax, perJoint),
      flag(_isTrqConOn), reals(_JntTrqMax, perJoint), reals(_JntTrqMin, perJoint), flag(_isCartVelConOn), real(_CartVelMax),
      flag(_isCartAccConOn), real(_CartAccMax),
      skipLines(2),
      real(_integRes), real(_maxIntegTime),
      skipLines(2),
      integer(_inputDecimFact), integer(_smoothWindow), flag(is_sdotOut), real(_jntThresh), real(_cartThresh), reals(_sWeights, 3),
      integer(_scaleType), real(_thetaNormRes), real(_thetaNormRes2), real(_cartNormRes), real(_cartNormRes2), real(_outRes),
      real(_outSmoothFact), flag(_isSVD), flag(_isPar2Ser),
   };
   int got = 0;
   for (const Field &f : layout) got += parseField(f, cur, _nJoints);

   _trajFileName = _InputFolder + trajName;
   _pathType = pathTypeFromString(_pathTypeStr);
   if (finishConfig() != 0) return -1;

   const int expected = 34 + 4 * (int)_nJoints;   // the item count the reference's scanner checks (ba.cpp:2075-2082)
   if (got != expected)
   {
      printf("\nfscanf error while reading config.dat file: returned %d; should be %d.\n", got, expected);
      return -1;
   }
   return 0;
}

int BA::loadConfigData(const Config &conf)
{
   _robotTypeStr = conf.robotTypeStr;
   _isParallelMech = conf.isParallelMech;
   _nJoints = conf.nJoints;
   _nCart = conf.nCart;
   _trajFileName = conf.trajFileName;
   _isBINfile = conf.isBinFile;
   _pathTypeStr = conf.pathType;
   _pathType = pathTypeFromString(_pathTypeStr);

   _isJntVelConOn = conf.isJntVelConon;   _JntVelMax = conf.jntVelLims;
   _isJntAccConOn = conf.isJntAccConOn;   _JntAccMax = conf.jntAccLims;
   _isTrqConOn = conf.isTrqConOn;         _JntTrqMax = conf.jntTrqMax;   _JntTrqMin = conf.jntTrqMin;
   _isCartVelConOn = conf.isCartVelConOn; _CartVelMax = conf.cartVelMax;
   _isCartAccConOn = conf.isCarAccConOn;  _CartAccMax = conf.cartAccMax;

   _integRes = conf.integRes;
   _maxIntegTime = conf.maxIntegTime;

   _inputDecimFact = conf.inputDecimFact;
   _smoothWindow = conf.smoothWindow;
   is_sdotOut = conf.is_sdotOut;
   _jntThresh = conf.jntThresh;
   _cartThresh = conf.cartThresh;
   _sWeights = conf.sWeights;
   _scaleType = conf.scaleType;
   _thetaNormRes = conf.thetaNormRes;   _thetaNormRes2 = conf.thetaNormRes2;
   _cartNormRes = conf.cartNormRes;     _cartNormRes2 = conf.cartNormRes2;
   _outRes = conf.outRes;
   _outSmoothFact = conf.outSmoothFact;
   _isSVD = conf.isSVD;
   _isPar2Ser = conf.isPar2Ser;
   return finishConfig();
}

// ---------------------------------------------------------------------------------------------
// trajectory input
// ---------------------------------------------------------------------------------------------
int BA::loadTrajectoryData(Traj &traj)
{
   traj.trajFileName = _trajFileName;
   for (std::vector<double> *v : {&traj.thetapt, &traj.thetaDpt, &traj.thetaD2pt}) v->resize(_nJoints);
   for (std::vector<double> *v : {&traj.cartpt, &traj.cartDpt, &traj.cartD2pt}) v->resize(_nCart);

   const char *fname = _trajFileName.c_str();
   if (doesFileExist(fname) != 0)
   {
      printf("Error: The file '%s' does not exist.\n", fname);
      return -1;
   }
   if ((_isBINfile ? trajReadBIN(traj, fname) : trajReadCSV(traj, fname)) == -1) return -1;
   printInputData(traj);
   return 0;
}

// float32 tres; int32 nPts; int32 hasTheta; [float32 theta[nJ][nPts]]; int32 hasCart; [float32 cart[nC][nPts]]
int BA::trajReadBIN(Traj &traj, const char *filename)
{
   FILE *fid = fopen(filename, "rb");
   if (fid == nullptr)
   {
      printf("\nError! Binary trajectory file '%s' doesn't exist!\n", filename);
      return -1;
   }
   PodReader in(fid);
   float res32 = 0;
   int present[2] = {0, 0};
   in.get(res32);
   in.get(traj.nPts);
   traj.tresInput = (double)res32;
   traj.sres = traj.tresInput;
   const size_t n = (size_t)traj.nPts;
   std::vector<std::vector<double>> *group[2] = {&traj.theta, &traj.cart};
   const unsigned int rows[2] = {_nJoints, _nCart};
   for (int g = 0; g < 2; ++g)
   {
      in.get(present[g]);
      if (present[g] != 1) continue;
      group[g]->resize(rows[g]);
      for (unsigned int j = 0; j < rows[g]; ++j) in.row((*group[g])[j], n);
   }
   fclose(fid);

   const int expected = (present[0] * (int)_nJoints + present[1] * (int)_nCart) * (int)traj.nPts + 4;
   if ((int)in.items() != expected)
   {
      printf("\nfread error: %d items read, %d items should have been read.\n", (int)in.items(), expected);
      return -1;
   }
   return 0;
}

// The header row names the columns ("timestamp", "j1" .., "x" ..); robots without kinematic model carry joints only.
// Rows are counted the way the reference counts them (ba.cpp:2345-2353): a row must start with a number and be
// terminated by a newline.
int BA::trajReadCSV(Traj &traj, const char *filename)
{
   NumericLocale pin;
   std::ifstream in(filename, std::ios::binary);
   if (!in)
   {
      printf("\nError! File %s doesn't exist", filename);
      return -1;
   }
   std::stringstream whole;
   whole << in.rdbuf();
   const std::string text = whole.str();

   // split into newline-terminated lines (a last line without terminator is not a row)
   std::vector<std::string> lines;
   for (size_t at = 0; at < text.size();)
   {
      const size_t nl = text.find('\n', at);
      if (nl == std::string::npos) break;
      lines.push_back(text.substr(at, nl - at));
      at = nl + 1;
   }
   auto cells = [](const std::string &line) {
      std::vector<std::string> out;
      std::string cell;
      std::istringstream ss(line);
      while (std::getline(ss, cell, ','))
      {
         const size_t a = cell.find_first_not_of(" \t\r"), b = cell.find_last_not_of(" \t\r");
         out.push_back(a == std::string::npos ? std::string() : cell.substr(a, b - a + 1));
      }
      return out;
   };

   const size_t nFields = _isGenericRobot ? _nJoints : _nJoints + _nCart + 1;
   traj.nPts = 0;
   std::vector<std::vector<std::string>> rows;
   for (size_t k = 1; k < lines.size(); ++k)
   {
      std::vector<std::string> c = cells(lines[k]);
      double first;
      if (c.empty() || !toDouble(c[0], first)) break;
      rows.push_back(c);
   }
   traj.nPts = (int)rows.size();
   if (traj.nPts == 0) return 0;

   size_t items = 0;
   bool hasTime = false, hasJoints = false, hasCart = false;
   const std::vector<std::string> head = lines.empty() ? std::vector<std::string>() : cells(lines[0]);
   traj.trajFileHeader.assign(nFields, std::string());
   for (size_t k = 0; k < nFields && k < head.size(); ++k)
   {
      if (head[k].empty()) break;
      traj.trajFileHeader[k] = head[k];
      ++items;
      hasTime = hasTime || head[k] == "timestamp";
      hasJoints = hasJoints || head[k] == "j1";
      hasCart = hasCart || head[k] == "x";
   }

   const size_t n = (size_t)traj.nPts;
   traj.timestamp.assign(n, 0.0);
   if (hasJoints) { traj.theta.resize(_nJoints); for (auto &ch : traj.theta) ch.resize(n, 0.0); }
   if (hasCart) { traj.cart.resize(_nCart); for (auto &ch : traj.cart) ch.resize(n, 0.0); }
   for (size_t i = 0; i < n; ++i)
   {
      const std::vector<std::string> &c = rows[i];
      size_t col = 0;
      auto take = [&](double &dst) {
         if (col < c.size() && toDouble(c[col], dst)) ++items;
         ++col;
      };
      if (hasTime) take(traj.timestamp[i]);
      if (hasJoints)
         for (size_t j = 0; j < _nJoints; ++j) take(traj.theta[j][i]);
      if (hasCart)
         for (size_t j = 0; j < _nCart; ++j) take(traj.cart[j][i]);
   }

   if (!hasTime)
      for (size_t i = 0; i < n; ++i) traj.timestamp[i] = 0.2 * (double)i;   // no timestamps: 0.2 s between rows (ba.cpp:2440-2444)
   traj.tresInput = traj.timestamp.back() / (traj.nPts - 1);
   traj.sres = traj.tresInput;

   if (nFields * (n + 1) != items)
   {
      printf("trajReadCSV: The number of items read from %s was %d. It should have been %d.\n", filename, (int)items,
             (int)(nFields * (n + 1)));
      printf("Most likely the run environment is not EN_US and fscanf is expecting commas for the decimal.\n");
      return -1;
   }
   return 0;
}

int BA::printInputData(const Traj &traj)
{
   auto list = [](const char *label, const std::vector<double> &v, unsigned int n) {
      printf("%s", label);
      for (unsigned int j = 0; j < n && j < v.size(); ++j) printf("%.1f ", v[j]);
      printf("\n");
   };
   printf("\n");
   printf("Robot: %s \n", _robotTypeStr.c_str());
   printf("Number of robot joints: %u \n", _nJoints);
   printf("Input  traj. file : %s\n", traj.trajFileName.c_str());
   printf("Input resolution  :  %.4f s\n", traj.tresInput);
   printf("Number of traj pts: %d\n", traj.nPts);
   list("Joint velocity limits : ", _JntVelMax, _nJoints);
   list("Joint accel.   limits : ", _JntAccMax, _nJoints);
   printf("Cartesian speed  limit: %.4f\n", _CartVelMax);
   printf("Integration resolution: %.4f s\n", _integRes);
   printf("Output      resolution: %.4f s\n", _outRes);
   printf("Max. integration time : %.0f s\n", _maxIntegTime);
   printf("\n");
   return 0;
}

// ---------------------------------------------------------------------------------------------
// output files
// ---------------------------------------------------------------------------------------------
int BA::writeOutputData(Traj &traj)
{
   trajWriteBIN(traj, (_OutputFolder + "traj_out.dat").c_str());
   if (!_isBINfile) trajWriteCSV(traj, (_OutputFolder + "traj_out.csv").c_str());
   if (is_sdotOut && !_isInterpOnly) sdotWrite(traj, (_OutputFolder + "s-sdot.dat").c_str());
   printf("\nOutput trajectory is %.3f sec.\n", (traj.nPts - 1) * traj.sres);
   return 0;
}

// float32 sres; int32 nPts; int32 1; float32 theta[nJ][nPts]; int32 hasCart; [cart]; int32 hasTrq; [trq]
int BA::trajWriteBIN(Traj &traj, const char *fname)
{
   FILE *fid = fopen(fname, "wb");
   if (fid == nullptr)
   {
      printf("\nUnable to open file %s", fname);
      return -1;
   }
   if (traj.theta.empty())
   {
      printf("trajWrite(): myTraj is empty; no file was written.\n");
      fclose(fid);
      return -1;
   }
   const size_t n = traj.theta[0].size();
   const int withTheta = 1;
   const int withCart = (traj.cart.size() == _nCart && !traj.cart.empty() && traj.cart[0].size() == n) ? 1 : 0;
   const int withTrq = (_isTrqConOn && !traj.trq.empty() && !traj.trq[0].empty()) ? 1 : 0;

   F32RowWriter out(fid);
   out.put((float)traj.sres);
   out.put(traj.nPts);
   out.put(withTheta);
   out.rows(traj.theta, _nJoints);
   out.put(withCart);
   if (withCart) out.rows(traj.cart, _nCart);
   out.put(withTrq);
   if (withTrq) out.rows(traj.trq, _nJoints);
   fclose(fid);
   return 0;
}

int BA::trajWriteCSV(Traj &traj, const char *fname)
{
   NumericLocale pin;
   FILE *fid = fopen(fname, "w");
   if (fid == nullptr)
   {
      printf("\nUnable to open file %s", fname);
      return -1;
   }
   for (size_t k = 0; k < traj.trajFileHeader.size(); ++k)
      fprintf(fid, k + 1 < traj.trajFileHeader.size() ? "%s, " : "%s\n", traj.trajFileHeader[k].c_str());

   if (traj.nPts != traj.timestamp.size()) _isInterpolated = true;
   const bool withCart = (traj.cart.size() == _nCart && !traj.cart.empty() && traj.cart[0].size() == traj.nPts);
   for (unsigned int i = 0; i < traj.nPts; ++i)
   {
      fprintf(fid, "%8.3f", _isInterpolated ? i * traj.sres : traj.timestamp[i]);
      for (unsigned int j = 0; j < _nJoints; ++j) fprintf(fid, ", %11.6f", traj.theta[j][i]);
      if (withCart)
         for (unsigned int j = 0; j < _nCart; ++j) fprintf(fid, ", %9.6f", traj.cart[j][i]);
      fprintf(fid, "\n");
   }
   fclose(fid);
   return 0;
}

// twice (reverse curve, forward curve): float64 sres; int32 n; float32 s[n]; float32 sdot[n]
int BA::sdotWrite(Traj &traj, const char *fname)
{
   FILE *fid = fopen(fname, "wb");
   if (fid == nullptr)
   {
      printf("\nUnable to open file %s", fname);
      return -1;
   }
   F32RowWriter out(fid);
   for (int k = 0; k < 2; ++k)
   {
      const int n = k < (int)traj.myMVChist.s.size() ? (int)traj.myMVChist.s[k].size() : 0;
      if (n <= 0)
      {
         printf("sdotWrite(): %s was not written because sdot is empty.\n", fname);
         fclose(fid);
         return -1;
      }
      out.put(traj.sres);
      out.put(n);
      out.row(traj.myMVChist.s[k]);
      out.row(traj.myMVChist.sdot[k]);
   }
   fclose(fid);
   return 0;
}

} // namespace BATOTP
