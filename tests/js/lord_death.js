// EmfLord.trainOnGpus must reject (not hang) when a per-GPU process dies from a signal.
'use strict';
const path = require('path');
const EmfLord = require(path.join(__dirname, '..', '..', 'you-can-not-recommend_amd', 'lib', 'emf', 'EmfLord.js'));
const lord = new EmfLord();
lord.init({}, { gpus: 2, commTransport: 'shm', gpuDevices: 1, gpuProcessScript: path.join(__dirname, 'dying_gpu_process.js'),
                gpuProcessTimeoutMs: Number(process.env.YCNR_TEST_TIMEOUT_MS || 0) });
const t0 = Date.now();
lord.trainOnGpus({ inline: { users: 1, items: 1, user: [0], item: [0], rating: [1] } }).then(
  () => { console.log(JSON.stringify({ outcome: 'resolved' })); process.exit(0); },
  (e) => { console.log(JSON.stringify({ outcome: 'rejected', error: String(e && e.message), ms: Date.now() - t0 })); process.exit(0); });
