// index.js — the JS library API of tidal-wave (reference: /root/reference/index.js:1-72), unchanged in shape:
// create(targetDir, {expectDir | getExpectedPath, ...engine options}) -> EventEmitter with 'data' / 'error' /
// 'finish'.  The reference walks the target directory with `glob-stream`; that module is not installed here,
// so the walk uses fs directly (same '**/*.*' semantics: every file whose name contains a dot).
'use strict';
var EventEmitter = require('events').EventEmitter,
  TidalWave = require('./build/Release/tidalwave').TidalWave,
  Path = require('path'),
  FS = require('fs');
Object.setPrototypeOf(TidalWave.prototype, EventEmitter.prototype);

module.exports.create = create;
module.exports.TidalWave = TidalWave;

function create(targetDir, options) {
  var t = new TidalWave(options);
  options = options || {};

  var getExpectedPath;
  if (typeof options.expectDir === 'string') {
    getExpectedPath = function(shortPath) {
      return Path.resolve(options.expectDir, shortPath);
    };
  } else if (typeof options.getExpectedPath === 'function') {
    getExpectedPath = options.getExpectedPath;
  } else {
    throw new Error('An option must have "expectDir" or "getExpectedPath" property.');
  }

  calcAll(t, targetDir, getExpectedPath);
  return t;
}

function walk(dir, out) {
  var entries;
  try { entries = FS.readdirSync(dir, { withFileTypes: true }); } catch (e) { return out; }
  entries.sort(function(a, b) { return a.name < b.name ? -1 : 1; });
  entries.forEach(function(e) {
    var p = Path.join(dir, e.name);
    if (e.isDirectory()) walk(p, out);
    else if (e.name.indexOf('.') !== -1) out.push(p);
  });
  return out;
}

function calcAll(tidalwave, targetDir, getExpectedPath) {
  var base = Path.resolve(targetDir);
  var files = walk(base, []);
  var globEnded = false;
  var requested = 0;
  var pending = files.length;

  if (files.length === 0) {
    // index.js:39-44 of the reference: nothing matched -> dispose at once -> 'finish' with zero counts
    setImmediate(function() { tidalwave.dispose(); });
    return;
  }

  files.forEach(function(target) {
    var shortPath = Path.relative(base, target);
    var expectedFile = getExpectedPath.call(this, shortPath);
    if (!expectedFile) { done(); return; }
    if (typeof expectedFile === 'string') calcIfExists(expectedFile);
    else if (typeof expectedFile.then === 'function') expectedFile.then(calcIfExists);
    function calcIfExists(expectedFile) {
      // the reference ignores the result of FS.exists (index.js:57-60): a missing expected file still
      // produces a calc and therefore an ERROR "Can't open ..."
      FS.access(expectedFile, function() {
        tidalwave.calc(expectedFile, target);
        requested++;
        done();
      });
    }
  });
  var disposed = false;
  function disposeIfIdle() {
    // every requested pair has answered and the walk is over: nothing else will arrive
    if (globEnded && requested <= 0 && !disposed) { disposed = true; tidalwave.dispose(); }
  }
  function done() {
    if (--pending === 0) {
      globEnded = true;
      disposeIfIdle();  // results that came in before the walk ended (or a walk that requested nothing)
    }
  }

  // The reference counts `requested` down only once the walk has ended (`globEnded && --requested <= 0`,
  // index.js:64-68) and, on 'error', never checks for completion (index.js:69-71): an event that arrives before the
  // last FS callback, or a run whose last event is an error, leaves the instance undisposed for ever.  With a
  // millisecond-latency engine both happen, so every event is counted and completion is checked at every step.
  tidalwave.on('data', function() { requested--; disposeIfIdle(); });
  tidalwave.on('error', function() { requested--; disposeIfIdle(); });
}
