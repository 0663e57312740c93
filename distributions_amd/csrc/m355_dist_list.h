// One line per distribution on the elementwise path: X(enum id, module name, number of parameters).
// Drives the launch declarations, the parameter-count table and the dispatch switches.
#pragma once
#define M355_DIST_LIST(X) \
  X(M355_NORMAL, normal, 2) \
  X(M355_GAMMA, gamma, 2) \
  X(M355_BETA, beta, 2) \
  X(M355_STUDENTT, studentt, 3) \
  X(M355_POISSON, poisson, 1) \
  X(M355_BINOMIAL, binomial, 2) \
  X(M355_EXPONENTIAL, exponential, 1) \
  X(M355_LOGNORMAL, lognormal, 2) \
  X(M355_HALFNORMAL, halfnormal, 1) \
  X(M355_CAUCHY, cauchy, 2) \
  X(M355_HALFCAUCHY, halfcauchy, 1) \
  X(M355_LAPLACE, laplace, 2) \
  X(M355_LOGISTIC, logistic, 2) \
  X(M355_GUMBEL, gumbel, 2) \
  X(M355_WEIBULL, weibull, 2) \
  X(M355_PARETO, pareto, 2) \
  X(M355_UNIFORM, uniform, 2) \
  X(M355_TRIANGULAR, triangular, 3) \
  X(M355_KUMARASWAMY, kumaraswamy, 2) \
  X(M355_MOYAL, moyal, 2) \
  X(M355_ASYMMETRICLAPLACE, asymmetriclaplace, 3) \
  X(M355_LOGLOGISTIC, loglogistic, 2) \
  X(M355_LOGITNORMAL, logitnormal, 2) \
  X(M355_INVERSEGAMMA, inversegamma, 2) \
  X(M355_CHISQUARED, chisquared, 1) \
  X(M355_SCALEDINVERSECHISQUARED, scaledinversechisquared, 2) \
  X(M355_HALFSTUDENTT, halfstudentt, 2) \
  X(M355_BETASCALED, betascaled, 4) \
  X(M355_BERNOULLI, bernoulli, 1) \
  X(M355_GEOMETRIC, geometric, 1) \
  X(M355_DISCRETEUNIFORM, discreteuniform, 2) \
  X(M355_DISCRETEWEIBULL, discreteweibull, 2) \
  X(M355_NEGATIVEBINOMIAL, negativebinomial, 2) \
  X(M355_BETABINOMIAL, betabinomial, 3) \
  X(M355_HYPERGEOMETRIC, hypergeometric, 3) \
  X(M355_ZI_POISSON, zi_poisson, 2) \
  X(M355_ZI_BINOMIAL, zi_binomial, 3) \
  X(M355_ZI_NEGATIVEBINOMIAL, zi_negativebinomial, 3) \
  X(M355_WALD, wald, 2) \
  X(M355_EXGAUSSIAN, exgaussian, 3) \
  X(M355_TRUNCATEDNORMAL, truncatednormal, 4) \
  X(M355_SKEW_STUDENTT, skew_studentt, 4) \
  X(M355_SKEWNORMAL, skewnormal, 3) \
  X(M355_VONMISES, vonmises, 2) \
  X(M355_RICE, rice, 2) \
  X(M355_POLYAGAMMA, polyagamma, 2)
