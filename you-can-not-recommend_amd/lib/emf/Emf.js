/**
 * Factory (lib/emf/Emf.js:30-59 of the reference). Only the Lord role exists here: one
 * process drives one GPU; Chief / Worker / Recommender belong to the reference's process
 * and TCP plumbing, which this path replaces.
 */
'use strict';

const EmfLord = require('./EmfLord');

class EmfFactory {
  static createLord() { return new EmfLord(); }
  static createWorker() { throw new Error('No worker processes: the portion loop runs on the GPU'); }
}

module.exports = EmfFactory;
