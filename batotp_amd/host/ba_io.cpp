// ba_io.cpp -- configuration and trajectory file IO of BA (host only).
//
// Restates reference batotp/ba.cpp:1942-2087 (readConfigData), 2100-2197 (loadConfigData),
// 2206-2245 (loadTrajectoryData), 2257-2312 (trajReadBIN), 2322-2461 (trajReadCSV), 2470-2501
// (printInputData), 2510-2528 (writeOutputData), 2582-2651 (trajWriteBIN), 2660-2717
// (trajWriteCSV) and 2726-2759 (sdotWrite).  File formats: SURVEY.md Appendix A.
#include <clocale>
#include <cstdio>
#include <numeric>

#include "ba.h"
#include "util.h"

namespace BATOTP
{

namespace
{
// the parsers expect '.' decimals: pin LC_NUMERIC while a file is open, restore afterwards
class NumericLocale
{
public:
   NumericLocale() : _saved(std::setlocale(LC_NUMERIC, NULL)) { std::setlocale(LC_NUMERIC, "en_US.UTF-8"); }
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
} // namespace

// ---------------------------------------------------------------------------------------------
// config.dat: 3 header lines, then one item per line in a fixed order
// ---------------------------------------------------------------------------------------------
int BA::readConfigData(const char *filename)
{
   NumericLocale pin;
   FILE *fid = fopen(filename, "r");
   if (fid == nullptr)
   {
      printf("\nUnable to open file %s\n", filename);
      return -1;
   }
   printf("\nConfiguration file: '%s'\n", filename);

   int got = 0; // number of items fscanf converted
   for (int k = 0; k < 3; ++k) NextLine(fid);

   _robotTypeStr = readChar(fid, got);
   _isParallelMech = readBool(fid, got);
   _isParallelMechOrig = _isParallelMech;
   _robotType = myRobot.call_set_robotType(_robotTypeStr);
   _isGenericRobot = (_robotTypeStr == "GENJNT");
   if (_robotType == 0)
   {
      fclose(fid);
      printf("\nreadInputData() error: robotType is %s", _robotTypeStr.c_str());
      printf("It should be 'KUKA', 'UR', 'RR', 'CSPR3DOF', or 'GENJNT'.\n");
      return -1;
   }
   _nJoints = readInt(fid, got);
   _nCart = readInt(fid, got);

   char name[FILENAME_MAX];
   got += fscanf(fid, "%s", name);
   _trajFileName = _InputFolder + name;
   NextLine(fid);
   _isBINfile = readBool(fid, got);

   const std::string pathTypeStr = readChar(fid, got);
   _pathType = pathTypeFromString(pathTypeStr);
   if (_pathType == 0)
   {
      fclose(fid);
      printf("\nreadInputData() error: pathType is %s", pathTypeStr.c_str());
      printf("It should be 'JOINT', 'CART', or 'BOTH'.\n");
      return -1;
   }
   NextLine(fid);
   NextLine(fid);

   // constraints
   _areJointAnglesDegrees = readBool(fid, got);
   _isJntVelConOn = readBool(fid, got);
   _JntVelMax = readDoubleVector(fid, got, _nJoints);
   _isJntAccConOn = readBool(fid, got);
   _JntAccMax = readDoubleVector(fid, got, _nJoints);
   _isTrqConOn = readBool(fid, got);
   _JntTrqMax = readDoubleVector(fid, got, _nJoints);
   _JntTrqMin = readDoubleVector(fid, got, _nJoints);
   for (unsigned int j = 0; j < _nJoints; ++j)
   {
      // NAN lower torque limit = symmetric limits
      if (std::isnan(_JntTrqMin[j])) _JntTrqMin[j] = -_JntTrqMax[j];
   }
   _isCartVelConOn = readBool(fid, got);
   _CartVelMax = readDouble(fid, got);
   _isCartAccConOn = readBool(fid, got);
   _CartAccMax = readDouble(fid, got);
   NextLine(fid);
   NextLine(fid);

   // integration
   _integRes = readDouble(fid, got);
   _maxIntegTime = readDouble(fid, got);
   NextLine(fid);
   NextLine(fid);

   // other controls
   _inputDecimFact = readInt(fid, got);
   _smoothWindow = readInt(fid, got);
   is_sdotOut = readBool(fid, got);
   _jntThresh = readDouble(fid, got);
   _cartThresh = readDouble(fid, got);
   _quadraticRadThresh = _cartThresh * _cartThresh;
   _sWeights = readDoubleVector(fid, got, 3);
   _scaleType = readInt(fid, got);
   _thetaNormRes = readDouble(fid, got);
   _thetaNormRes2 = readDouble(fid, got);
   _cartNormRes = readDouble(fid, got);
   _cartNormRes2 = readDouble(fid, got);
   _outRes = readDouble(fid, got);
   _outSmoothFact = readDouble(fid, got);
   _isSVD = readBool(fid, got);
   _isPar2Ser = readBool(fid, got);
   fclose(fid);
   if (_isSVD && _isParallelMech && _isTrqConOn)
   {
      // reference util.cpp:421-438 solves the wrench system with Eigen's JacobiSVD when isSVD = 1; only the LU solve
      // (isSVD = 0, every shipped configuration) exists here -- refuse rather than answer with a different solver
      printf("Error in readInputData(): isSVD = 1 (Jacobi-SVD solve of the cable wrench system) is not implemented; use isSVD = 0 (LU).\n");
      return -1;
   }

   const double wSum = _sWeights[0] + _sWeights[1] + _sWeights[2];
   if (wSum <= 0)
   {
      printf("Error in readInputData(): sum(sWeights) should be greater than 0.\n");
      return -1;
   }
   for (int k = 0; k < 3; ++k) _sWeights[k] /= wSum;

   const int expected = 34 + 4 * _nJoints;
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
   _isParallelMechOrig = _isParallelMech;
   _robotType = myRobot.call_set_robotType(_robotTypeStr);
   _isGenericRobot = (_robotTypeStr == "GENJNT");
   if (_robotType == 0)
   {
      printf("\nreadInputData() error: robotType is %s", _robotTypeStr.c_str());
      printf("It should be 'KUKA', 'UR', 'RR', 'CSPR3DOF', or 'GENJNT'.\n");
      return -1;
   }
   _nJoints = conf.nJoints;
   _nCart = conf.nCart;
   _trajFileName = conf.trajFileName;
   _isBINfile = conf.isBinFile;
   _pathType = pathTypeFromString(conf.pathType);
   if (_pathType == 0)
   {
      printf("\nreadInputData() error: pathType is %s", conf.pathType.c_str());
      printf("It should be 'JOINT', 'CART', or 'BOTH'.\n");
      return -1;
   }

   _isJntVelConOn = conf.isJntVelConon;
   _JntVelMax = conf.jntVelLims;
   _isJntAccConOn = conf.isJntAccConOn;
   _JntAccMax = conf.jntAccLims;
   _isTrqConOn = conf.isTrqConOn;
   _JntTrqMax = conf.jntTrqMax;
   _JntTrqMin = conf.jntTrqMin;
   for (unsigned int j = 0; j < _nJoints; ++j)
   {
      if (std::isnan(_JntTrqMin[j])) _JntTrqMin[j] = -_JntTrqMax[j];
   }
   _isCartVelConOn = conf.isCartVelConOn;
   _CartVelMax = conf.cartVelMax;
   _isCartAccConOn = conf.isCarAccConOn;
   _CartAccMax = conf.cartAccMax;

   _integRes = conf.integRes;
   _maxIntegTime = conf.maxIntegTime;

   _inputDecimFact = conf.inputDecimFact;
   _smoothWindow = conf.smoothWindow;
   is_sdotOut = conf.is_sdotOut;
   _jntThresh = conf.jntThresh;
   _cartThresh = conf.cartThresh;
   _quadraticRadThresh = _cartThresh * _cartThresh;
   _sWeights = conf.sWeights;
   _scaleType = conf.scaleType;
   _thetaNormRes = conf.thetaNormRes;
   _thetaNormRes2 = conf.thetaNormRes2;
   _cartNormRes = conf.cartNormRes;
   _cartNormRes2 = conf.cartNormRes2;
   _outRes = conf.outRes;
   _outSmoothFact = conf.outSmoothFact;
   _isSVD = conf.isSVD;
   _isPar2Ser = conf.isPar2Ser;
   if (_isSVD && _isParallelMech && _isTrqConOn)
   {
      // reference util.cpp:421-438 solves the wrench system with Eigen's JacobiSVD when isSVD = 1; only the LU solve
      // (isSVD = 0, every shipped configuration) exists here -- refuse rather than answer with a different solver
      printf("Error in readInputData(): isSVD = 1 (Jacobi-SVD solve of the cable wrench system) is not implemented; use isSVD = 0 (LU).\n");
      return -1;
   }

   const double wSum = _sWeights[0] + _sWeights[1] + _sWeights[2];
   if (wSum <= 0)
   {
      printf("Error in readInputData(): sum(sWeights) should be greater than 0.\n");
      return -1;
   }
   for (int k = 0; k < 3; ++k) _sWeights[k] /= wSum;
   return 0;
}

// ---------------------------------------------------------------------------------------------
// trajectory input
// ---------------------------------------------------------------------------------------------
int BA::loadTrajectoryData(Traj &traj)
{
   traj.trajFileName = _trajFileName;
   traj.thetapt.resize(_nJoints);
   traj.thetaDpt.resize(_nJoints);
   traj.thetaD2pt.resize(_nJoints);
   traj.cartpt.resize(_nCart);
   traj.cartDpt.resize(_nCart);
   traj.cartD2pt.resize(_nCart);

   const char *fname = _trajFileName.c_str();
   if (doesFileExist(fname) != 0)
   {
      printf("Error: The file '%s' does not exist.\n", fname);
      return -1;
   }
   const int rc = _isBINfile ? trajReadBIN(traj, fname) : trajReadCSV(traj, fname);
   if (rc == -1) return -1;
   printInputData(traj);
   return 0;
}

// float32 tres; int32 nPts; int32 hasTheta; [float32 theta[nJ][nPts]]; int32 hasCart;
// [float32 cart[nC][nPts]]
int BA::trajReadBIN(Traj &traj, const char *filename)
{
   FILE *fid = fopen(filename, "rb");
   if (fid == nullptr)
   {
      printf("\nError! Binary trajectory file '%s' doesn't exist!\n", filename);
      return -1;
   }
   size_t items = 0;
   float res32 = 0;
   int hasTheta = 0, hasCart = 0;

   items += fread(&res32, 4, 1, fid);
   traj.tresInput = (double)res32;
   traj.sres = traj.tresInput;
   items += fread(&traj.nPts, 4, 1, fid);
   std::vector<float> row(traj.nPts);

   items += fread(&hasTheta, 4, 1, fid);
   if (hasTheta == 1)
   {
      traj.theta.resize(_nJoints, std::vector<double>(traj.nPts));
      for (unsigned int j = 0; j < _nJoints; ++j)
      {
         items += fread(row.data(), 4, traj.nPts, fid);
         std::copy(row.begin(), row.end(), traj.theta[j].begin());
      }
   }
   items += fread(&hasCart, 4, 1, fid);
   if (hasCart == 1)
   {
      traj.cart.resize(_nCart, std::vector<double>(traj.nPts));
      for (unsigned int j = 0; j < _nCart; ++j)
      {
         items += fread(row.data(), 4, traj.nPts, fid);
         std::copy(row.begin(), row.end(), traj.cart[j].begin());
      }
   }
   fclose(fid);

   const int expected = (hasTheta * _nJoints + hasCart * _nCart) * traj.nPts + 4;
   if ((int)items != expected)
   {
      printf("\nfread error: %d items read, %d items should have been read.\n", (int)items, expected);
      return -1;
   }
   return 0;
}

// header line names the columns ("timestamp", "j1".., "x"..); generic robots carry joints only
int BA::trajReadCSV(Traj &traj, const char *filename)
{
   NumericLocale pin;
   FILE *fid = fopen(filename, "r");
   if (fid == nullptr)
   {
      printf("\nError! File %s doesn't exist", filename);
      return -1;
   }
   const size_t nFields = _isGenericRobot ? _nJoints : _nJoints + _nCart + 1;
   traj.trajFileHeader.resize(nFields);

   // count data rows
   NextLine(fid);
   traj.nPts = 0;
   for (;;)
   {
      double first;
      if (fscanf(fid, "%lf", &first) != 1) break;
      if (NextLine(fid) == EOF) break;
      traj.nPts++;
   }
   if (traj.nPts == 0) return 0;
   rewind(fid);

   bool hasTime = false, hasJoints = false, hasCart = false;
   size_t items = 0;
   char word[100];
   for (size_t k = 0; k < nFields; ++k)
   {
      items += fscanf(fid, " %99[^, \t\n],", word);
      traj.trajFileHeader[k] = word;
      if (traj.trajFileHeader[k] == "timestamp") hasTime = true;
      if (traj.trajFileHeader[k] == "j1") hasJoints = true;
      if (traj.trajFileHeader[k] == "x") hasCart = true;
   }

   traj.timestamp.resize(traj.nPts);
   if (hasJoints)
   {
      traj.theta.resize(_nJoints, std::vector<double>(traj.nPts));
      for (auto &ch : traj.theta) ch.resize(traj.nPts, 0);
   }
   if (hasCart)
   {
      traj.cart.resize(_nCart, std::vector<double>(traj.nPts));
      for (auto &ch : traj.cart) ch.resize(traj.nPts, 0);
   }
   for (size_t i = 0; i < traj.nPts; ++i)
   {
      if (hasTime) items += fscanf(fid, "%lf,", &traj.timestamp[i]);
      if (hasJoints)
         for (size_t j = 0; j < _nJoints; ++j) items += fscanf(fid, "%lf,", &traj.theta[j][i]);
      if (hasCart)
         for (size_t j = 0; j < _nCart; ++j) items += fscanf(fid, "%lf,", &traj.cart[j][i]);
   }
   fclose(fid);

   if (!hasTime)
   {
      // no timestamps: assume 0.2 s between rows
      for (size_t i = 0; i < traj.timestamp.size(); ++i) traj.timestamp[i] = 0.2 * (double)i;
   }
   traj.tresInput = traj.timestamp.back() / (traj.nPts - 1);
   traj.sres = traj.tresInput;

   if (nFields * (traj.nPts + 1) != items)
   {
      printf("trajReadCSV: The number of items read from %s was %d. It should have been %d.\n", filename,
             (int)items, (int)nFields * (traj.nPts + 1));
      printf("Most likely the run environment is not EN_US and fscanf is expecting commas for the decimal.\n");
      return -1;
   }
   return 0;
}

int BA::printInputData(const Traj &traj)
{
   printf("\n");
   printf("Robot: %s \n", _robotTypeStr.c_str());
   printf("Number of robot joints: %u \n", _nJoints);
   printf("Input  traj. file : %s\n", traj.trajFileName.c_str());
   printf("Input resolution  :  %.4f s\n", traj.tresInput);
   printf("Number of traj pts: %d\n", traj.nPts);
   printf("Joint velocity limits : ");
   for (unsigned int j = 0; j < _nJoints; ++j) printf("%.1f ", _JntVelMax[j]);
   printf("\n");
   printf("Joint accel.   limits : ");
   for (unsigned int j = 0; j < _nJoints; ++j) printf("%.1f ", _JntAccMax[j]);
   printf("\n");
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
   std::string filename = _OutputFolder + "traj_out.dat";
   trajWriteBIN(traj, filename.c_str());
   if (!_isBINfile)
   {
      filename = _OutputFolder + "traj_out.csv";
      trajWriteCSV(traj, filename.c_str());
   }
   if (is_sdotOut && !_isInterpOnly)
   {
      filename = _OutputFolder + "s-sdot.dat";
      sdotWrite(traj, filename.c_str());
   }
   printf("\nOutput trajectory is %.3f sec.\n", (traj.nPts - 1) * traj.sres);
   return 0;
}

// float32 sres; int32 nPts; int32 1; float32 theta[nJ][nPts]; int32 hasCart; [cart];
// int32 hasTrq; [trq]
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
      return -1;
   }
   const size_t n = traj.theta[0].size();
   int hasTheta = 1, hasCart = 0, hasTrq = 0;
   if (traj.cart.size() == _nCart && traj.cart[0].size() == n) hasCart = 1;
   if (_isTrqConOn && !traj.trq.empty() && !traj.trq[0].empty()) hasTrq = 1;

   const float res32 = (float)traj.sres;
   fwrite(&res32, 4, 1, fid);
   fwrite(&traj.nPts, 4, 1, fid);
   fwrite(&hasTheta, 4, 1, fid);

   std::vector<float> row(n);
   for (size_t j = 0; j < _nJoints; ++j)
   {
      std::copy(traj.theta[j].begin(), traj.theta[j].end(), row.begin());
      fwrite(row.data(), 4, n, fid);
   }
   fwrite(&hasCart, 4, 1, fid);
   if (hasCart == 1)
   {
      for (size_t j = 0; j < _nCart; ++j)
      {
         std::copy(traj.cart[j].begin(), traj.cart[j].end(), row.begin());
         fwrite(row.data(), 4, n, fid);
      }
   }
   fwrite(&hasTrq, 4, 1, fid);
   if (hasTrq)
   {
      for (size_t j = 0; j < _nJoints; ++j)
      {
         std::copy(traj.trq[j].begin(), traj.trq[j].end(), row.begin());
         fwrite(row.data(), 4, n, fid);
      }
   }
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
   for (unsigned int k = 0; k + 1 < traj.trajFileHeader.size(); ++k) fprintf(fid, "%s, ", traj.trajFileHeader[k].c_str());
   fprintf(fid, "%s\n", traj.trajFileHeader[traj.trajFileHeader.size() - 1].c_str());

   if (traj.nPts != traj.timestamp.size()) _isInterpolated = true;
   const bool hasCart = (traj.cart.size() == _nCart && traj.cart[0].size() == traj.nPts);

   for (unsigned int i = 0; i < traj.nPts; ++i)
   {
      if (_isInterpolated) fprintf(fid, "%8.3f", i * traj.sres);
      else fprintf(fid, "%8.3f", traj.timestamp[i]);
      for (unsigned int j = 0; j < _nJoints; ++j) fprintf(fid, ", %11.6f", traj.theta[j][i]);
      if (hasCart)
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
   for (int k = 0; k < 2; ++k)
   {
      const int n = (int)traj.myMVChist.s[k].size();
      if (n <= 0)
      {
         printf("sdotWrite(): %s was not written because sdot is empty.\n", fname);
         return -1;
      }
      std::vector<float> row(n);
      fwrite(&traj.sres, 8, 1, fid);
      fwrite(&n, 4, 1, fid);
      std::copy(traj.myMVChist.s[k].begin(), traj.myMVChist.s[k].end(), row.begin());
      fwrite(row.data(), 4, n, fid);
      std::copy(traj.myMVChist.sdot[k].begin(), traj.myMVChist.sdot[k].end(), row.begin());
      fwrite(row.data(), 4, n, fid);
   }
   fclose(fid);
   return 0;
}

} // namespace BATOTP
